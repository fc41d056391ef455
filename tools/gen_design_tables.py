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
    for name, f in (("C2 indy7 N=32 B=1024 (headline)", "%s_bench.json"), ("C3 iiwa14 N=128 B=256", "%s_bench_c3.json"), ("C5 shard iiwa14 N=64 B=512, sweep", "%s_bench_c5.json")):
        j = load(os.path.join(ROOT, "profiles", f % tag))
        if not j:
            continue
        r, c = j["roofline"], j.get("cpu_baseline", {})
        rows.append("| %s | %.3g | %.3f | %s | %.1f | %.3f (%s) | %s |" % (name, j["value"], j["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["bound"],
                                                                         "%.3g on %d cores" % (c["value"], c["cores"]) if c else "–"))
    return "Source: `profiles/%s_bench*.json` (the JSON lines of `bench.py`).\n\n" % tag + "\n".join(rows) + "\n"


def heatmap_block():
    """the reference's MPC solve-time heat-map next to this library's (profiles/r03_mpc_heatmap_*.json, tools/mpc_heatmap.py)"""
    cells, pub = {}, {}
    for f in ("short_pcg", "long_pcg", "long_direct"):
        path = os.path.join(ROOT, "profiles", "r04_mpc_heatmap_%s.json" % f)   # the latest round that measured it
        if not os.path.exists(path):
            path = os.path.join(ROOT, "profiles", "r03_mpc_heatmap_%s.json" % f)
        for c in json.load(open(path)):
            cells[(c["knots"], c["linear_solver"], c["batch"])] = c
            if c.get("published_ms") is not None:
                pub[(c["knots"], c["batch"])] = c["published_ms"]
    batches = sorted({k[2] for k in cells})
    rows = ["| N \\\\ batch | " + " | ".join(str(b) for b in batches) + " |", "|" + "---|" * (len(batches) + 1)]
    ahead = {"pcg": 0, "best": 0}
    for N in sorted({k[0] for k in cells}):
        for ls in ("pcg", "direct"):
            if (N, ls, batches[0]) not in cells:
                continue
            rows.append("| %d: this library, %s (ms) | " % (N, "PCG" if ls == "pcg" else "**direct**") +
                        " | ".join("%.3f" % cells[(N, ls, b)]["mean_ms"] if (N, ls, b) in cells else "–" for b in batches) + " |")
        rows.append("| %d: reference, published (ms) | " % N + " | ".join("%.2f" % pub[(N, b)] if (N, b) in pub else "–" for b in batches) + " |")
        for b in batches:
            if (N, b) in pub:
                ahead["pcg"] += cells[(N, "pcg", b)]["mean_ms"] < pub[(N, b)]
                ahead["best"] += min(cells[(N, ls, b)]["mean_ms"] for ls in ("pcg", "direct") if (N, ls, b) in cells) < pub[(N, b)]
    w = {(N, b): cells[(N, "pcg", b)]["step_wall_mean_ms"] for (N, ls, b) in cells if ls == "pcg"}
    tail = ("\nBy the device time of the solve, %d of the %d published cells are faster here with PCG and %d with the better of PCG and the direct mode; "
            "host wall time of the whole session step (`step_wall_mean_ms`): %.2f ms at N = 32, B = 1, %.2f ms at B = 256.\n"
            % (ahead["pcg"], len(pub), ahead["best"], w[(32, 1)], w[(32, 256)]))
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
