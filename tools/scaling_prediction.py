#!/usr/bin/env python3
"""A weak-scaling PREDICTION for the 8-GPU configurations, built from measurements on ONE MI355X (the driver's 8-GPU run has never been available).

The ranks of a sharded job hold different rows (C4: rows [r B, (r+1) B) of the 8192-row fig-8 batch; C5: shard r of the sweep), so their PCG
iteration counts -- and with them their solve times -- differ, and the node runs at the slowest rank's pace.  What a rank adds to the plain loop
when it is part of an N > 1 job is the library's sharded call sequence: snapshot, speculative solve, ONE ncclAllReduce of the count vector, the
host wait for the verdict, ONE ncclAllGather of the packed results on the communication stream.  Both are measurable on one device:

    for r in 0..7:  bench.py --as-rank r --of 8 [--one-rank-comm]      (200 timed solves each, the driver's own script)

    t_r(plain)   the rows of rank r through the single-GPU loop
    t_r(comm)    the same rows through the library's own communicator with world size 1 (RCCL's launch path, no wire)

    predicted efficiency at G ranks = t_0(plain) / max_{r < G} t_r(comm)
        numerator   = what the driver measures at N = 1 (rank 0's rows, no communicator)
        denominator = the slowest of the G ranks, each paying the sharded call sequence

What the prediction does NOT contain: the wire time of the two collectives between devices (40 bytes and 2.3 MB per rank and solve over xGMI:
~15 us for the gather at the per-link rate, overlapped with the next solve), RCCL's kernels competing for CUs with a solve on the SAME device
(the gather of solve n runs beside solve n+1), and launch skew between 8 host processes.  It is an upper bound built from hardware numbers, with
the attribution (skew of the shards | the sharded call sequence) split out.

    python tools/scaling_prediction.py [--steps 200] [--out gpurun_out/r06_scaling_prediction.json]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = {
    "C4": {"args": [], "what": "indy7 N=32, 1024 rows per rank of the 8192-row fig-8 batch (BASELINE config 4; ranks 0-1 / 0-3 are the 2- and 4-GPU jobs)"},
    "C5": {"args": ["--workload", "hparam", "--plant", "iiwa14", "--knots", "64", "--batch", "512"],
           "what": "iiwa14 N=64, 512 rows per rank = shard r of the hyper-parameter sweep (BASELINE config 5)"},
}


def run(extra, steps, warmup, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        raise RuntimeError("bench.py %s failed (rc %d): %s" % (" ".join(extra), r.returncode, r.stderr[-1500:]))
    return json.loads(lines[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--workloads", default="C4,C5")
    ap.add_argument("--periter", action="store_true", help="also time the per-iteration count mode (GATO_SOLVED_COUNT=periter) through the one-rank communicator")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_scaling_prediction.json"))
    a = ap.parse_args()
    out = {"what": __doc__.split("\n\n")[0], "steps": a.steps, "warmup": a.warmup, "workloads": {}}
    for name in a.workloads.split(","):
        w = WORKLOADS[name]
        rows = []
        for r in range(a.ranks):
            rec = {"rank": r}
            for key, extra, env in (("plain", [], None), ("comm", ["--one-rank-comm"], None)) + \
                    ((("comm_periter", ["--one-rank-comm"], dict(os.environ, GATO_SOLVED_COUNT="periter")),) if a.periter else ()):
                d = run(w["args"] + ["--as-rank", str(r), "--of", str(a.ranks)] + extra, a.steps, a.warmup, env)
                assert d["solution_ok"], (name, r, key, d["solution_checks"])
                rec[key] = {"ms_per_solve": d["ms_per_step"], "value": d["value"], "mean_pcg_iters": d["config"]["mean_pcg_iters"],
                            "sum_over_launches_of_max_pcg_iters": d["config"]["sum_over_launches_of_max_pcg_iters"],
                            "pcg_launch_us": d["roofline"]["avg_launch_us"], "stage_us_per_solve": d["roofline"]["stage_us_per_solve"],
                            "library": d["library"]["built_from"]}
                if "multi_gpu" in d:
                    mg = d["multi_gpu"]
                    rec[key].update({"gather_ms": mg["gather_ms"]["max_over_ranks"], "solve_ms_without_gather": mg["solve_ms_without_gather"],
                                     "solves_by_count_form": mg["solves_by_count_form"], "solved_count": mg["solved_count"]})
                print("%s rank %d %-12s %.4f ms per solve  (sum of the launches' max PCG iterations %d)" % (name, r, key, d["ms_per_step"], rec[key]["sum_over_launches_of_max_pcg_iters"]),
                      file=sys.stderr, flush=True)
            rows.append(rec)
        t0 = rows[0]["plain"]["ms_per_solve"]
        pred = {}
        for G in (2, 4, 8):
            if G > a.ranks:
                continue
            sub = rows[:G]
            slow_plain = max(r["plain"]["ms_per_solve"] for r in sub)
            slow_comm = max(r["comm"]["ms_per_solve"] for r in sub)
            pred[str(G)] = {"predicted_efficiency": t0 / slow_comm,
                            "from_shard_skew_alone": t0 / slow_plain,                                  # the slowest shard's rows, no communicator
                            "from_the_sharded_call_sequence_alone": rows[0]["plain"]["ms_per_solve"] / rows[0]["comm"]["ms_per_solve"],   # rank 0's rows, with / without
                            "slowest_rank": int(max(range(G), key=lambda i: sub[i]["comm"]["ms_per_solve"])),
                            "predicted_value": G * (1024 if name == "C4" else 512) * 10 / (slow_comm * 1e-3)}
        comm_cost = [r["comm"]["ms_per_solve"] - r["plain"]["ms_per_solve"] for r in rows]
        out["workloads"][name] = {"what": w["what"], "per_rank": rows, "t_rank0_plain_ms": t0, "prediction": pred,
                                  "sharded_call_sequence_cost_ms": {"mean": sum(comm_cost) / len(comm_cost), "min": min(comm_cost), "max": max(comm_cost)}}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["prediction"] for k, v in out["workloads"].items()}))


if __name__ == "__main__":
    main()
