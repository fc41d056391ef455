#!/usr/bin/env python3
"""Regenerates the measured tables of DESIGN.md from profiles/ so that prose and profiles cannot drift apart (round-2 verdict, weak #10):
the block between `<!-- BEGIN GENERATED <name> -->` and `<!-- END GENERATED <name> -->` is replaced.

    python tools/gen_design_tables.py r03a        # tag of the profiling round (profiles/<tag>_bench*.json, profiles/pmc_summary.json)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TITLES = {"c2": "C2 — indy7 N=32 B=1024 (the bench line)", "c3": "C3 — iiwa14 N=128 B=256", "c5": "C5 shard — iiwa14 N=64 B=512, sweep settings",
          "cr": "direct mode — indy7 N=128 B=8, one SQP iteration (block cyclic reduction)"}


def load(path):
    try:
        txt = open(path).read().strip().splitlines()
        return json.loads([l for l in txt if l.startswith("{")][-1])
    except Exception:
        return None


def kernels_block(pmc):
    out = ["Source: `profiles/pmc_summary.json` (tag `%s`, kernel sources `%s`, git `%s`); µs = `rocprofv3 --kernel-trace --stats` average, HBM = FETCH_SIZE / "
           "WRITE_SIZE counter passes (read side doubled per the gfx950 correction: an upper estimate), issue = SQ_INSTS_VALU × 4 cycles ÷ (duration × "
           "2.4 GHz × 1024 SIMDs), wait = SQ_WAIT_ANY ÷ SQ_WAVE_CYCLES." % (pmc.get("tag"), pmc.get("build"), pmc.get("git_head")), ""]
    for cfg in ("c2", "c3", "c5", "cr"):
        if cfg not in pmc:
            continue
        out += ["**%s**" % TITLES[cfg], "", "| kernel | launches | avg µs | min … max µs | % of GPU time | HBM MB / launch | VALU issue | wait | LDS conflicts | MFMA busy cycles |",
                "|---|---|---|---|---|---|---|---|---|---|"]
        for k, d in sorted(pmc[cfg].items(), key=lambda kv: -kv[1].get("pct_of_gpu_time", 0)):
            wait = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"] if d.get("SQ_WAVE_CYCLES") else None
            out.append("| `%s` | %d | %.1f | %.1f … %.1f | %.1f | %s | %s | %s | %s | %s |" % (
                k, d["calls"], d["avg_us"], d["min_us"], d["max_us"], d["pct_of_gpu_time"],
                "%.1f" % (d["hbm_bytes"] / 1e6) if "hbm_bytes" in d else "–", "%.2f" % d["valu_issue_frac"] if "valu_issue_frac" in d else "–",
                "%.2f" % wait if wait is not None else "–", "%.3f" % d["lds_bank_conflict_frac"] if "lds_bank_conflict_frac" in d else "–",
                "%.3g" % d["mfma_busy_cycles"] if "mfma_busy_cycles" in d else "–"))
        out.append("")
    return "\n".join(out)


def bench_block(tag):
    rows = ["| configuration | traj-SQP-iter/s | ms per solve | dominant kernel | µs / launch | roof (bound) | CPU port, host cores |", "|---|---|---|---|---|---|---|"]
    for name, f in (("C2 indy7 N=32 B=1024 (headline)", "%s_bench.json"), ("C3 iiwa14 N=128 B=256", "%s_bench_c3.json"), ("C5 shard iiwa14 N=64 B=512, sweep", "%s_bench_c5.json"),
                    ("one device at the C4 global batch: indy7 N=32 B=8192", "%s_bench_b8192.json"), ("C2 as the driver runs it (20 timed solves)", "%s_bench_driver_style.json")):
        j = load(os.path.join(ROOT, "profiles", f % tag))
        if not j:
            continue
        r, c = j["roofline"], j.get("cpu_baseline", {})
        rows.append("| %s | %.3g | %.3f | %s | %.1f | %.3f (%s) | %s |" % (name, j["value"], j["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["bound"],
                                                                         "%.3g on %d cores" % (c["value"], c["cores"]) if c else "–"))
    return "Source: `profiles/%s_bench*.json` (the JSON lines of `bench.py`).\n\n" % tag + "\n".join(rows) + "\n"


def heatmap_block():
    """the reference's MPC solve-time heat-map next to this library's (profiles/r06i_mpc_heatmap_{pcg,direct}.json, tools/mpc_heatmap.py --solve-wall).
    THREE figures per cell: the host wall clock around the solve alone, device-synchronised on both sides -- the reference's own `sqp_time_us`
    (bsqp.cuh:109,185), which is what its published heat-map shows, and therefore the column compared with it --, the device time between hipEvents
    around the solve's launches, and the host wall clock of the whole session step."""
    cells, pub = {}, {}
    for f in ("pcg", "direct"):
        path = os.path.join(ROOT, "profiles", "r06i_mpc_heatmap_%s.json" % f)      # the round's final kernels
        if not os.path.exists(path):
            path = os.path.join(ROOT, "profiles", "r06_mpc_heatmap_%s.json" % f)   # the first measurement of the round (before the PCG chain work)
        for c in json.load(open(path)):
            cells[(c["knots"], c["linear_solver"], c["batch"])] = c
            if c.get("published_ms") is not None:
                pub[(c["knots"], c["batch"])] = c["published_ms"]
    batches = sorted({k[2] for k in cells})
    rows = ["| N \\ batch | " + " | ".join(str(b) for b in batches) + " |", "|" + "---|" * (len(batches) + 1)]
    ahead = {("pcg", "solve_wall_mean_ms"): 0, ("pcg", "mean_ms"): 0, ("best", "solve_wall_mean_ms"): 0, ("best", "mean_ms"): 0, ("pcg", "step_wall_mean_ms"): 0, ("best", "step_wall_mean_ms"): 0}
    lost = []
    for N in sorted({k[0] for k in cells}):
        for ls in ("pcg", "direct"):
            if (N, ls, batches[0]) not in cells:
                continue
            name = "PCG" if ls == "pcg" else "**direct**"
            rows.append("| %d: %s, solve wall = `sqp_time_us` (ms) | " % (N, name) +
                        " | ".join(("**%.3f**" if (N, b) in pub and cells[(N, ls, b)]["solve_wall_mean_ms"] >= pub[(N, b)] else "%.3f") % cells[(N, ls, b)]["solve_wall_mean_ms"]
                                   if (N, ls, b) in cells else "–" for b in batches) + " |")
            rows.append("| %d: %s, device events (ms) | " % (N, name) + " | ".join("%.3f" % cells[(N, ls, b)]["mean_ms"] if (N, ls, b) in cells else "–" for b in batches) + " |")
        rows.append("| %d: whole session step, host wall, PCG (ms) | " % N + " | ".join("%.3f" % cells[(N, "pcg", b)]["step_wall_mean_ms"] for b in batches) + " |")
        rows.append("| %d: reference, published (ms) | " % N + " | ".join("%.2f" % pub[(N, b)] if (N, b) in pub else "–" for b in batches) + " |")
        for b in batches:
            if (N, b) not in pub:
                continue
            for key in ("solve_wall_mean_ms", "mean_ms", "step_wall_mean_ms"):
                ahead[("pcg", key)] += cells[(N, "pcg", b)][key] < pub[(N, b)]
                ahead[("best", key)] += min(cells[(N, ls, b)][key] for ls in ("pcg", "direct") if (N, ls, b) in cells) < pub[(N, b)]
            if cells[(N, "pcg", b)]["solve_wall_mean_ms"] >= pub[(N, b)]:
                lost.append("N = %d B = %d (%.3f vs %.2f)" % (N, b, cells[(N, "pcg", b)]["solve_wall_mean_ms"], pub[(N, b)]))
    n = len(pub)
    tail = ("\nBy the reference's own metric (host wall clock of the solve, bold = not ahead of the published cell): **%d of the %d published cells are faster here with PCG, "
            "%d with the better of PCG and the direct mode**; by device time %d / %d; by the wall clock of the WHOLE session step (plant, prepare, solve, selection, read-back) "
            "%d / %d.  PCG cells not ahead: %s.\n"
            % (ahead[("pcg", "solve_wall_mean_ms")], n, ahead[("best", "solve_wall_mean_ms")], ahead[("pcg", "mean_ms")], ahead[("best", "mean_ms")],
               ahead[("pcg", "step_wall_mean_ms")], ahead[("best", "step_wall_mean_ms")], "; ".join(lost) if lost else "none"))
    return "\n".join(rows) + "\n" + tail


def main(tag):
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    blocks = {"kernels": kernels_block(pmc), "bench": bench_block(tag), "heatmap": heatmap_block()}
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    for name, body in blocks.items():
        pat = re.compile(r"(<!-- BEGIN GENERATED %s -->\n).*?(<!-- END GENERATED %s -->)" % (name, name), re.S)
        if not pat.search(s):
            print("DESIGN.md has no block", name)
            continue
        s = pat.sub(lambda m: m.group(1) + body + "\n" + m.group(2), s)
    open(p, "w").write(s)
    print("DESIGN.md: regenerated", ", ".join(blocks))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03b")
