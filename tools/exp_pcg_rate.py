#!/usr/bin/env python3
"""Experiment: duration of the PCG launch as a function of its (maximum) iteration count, for a lone trajectory and for batches: run under
rocprofv3 --kernel-trace (writes the iteration counts of every launch to $OUT/iters.json), then `--parse <dir>` fits duration = a + b K."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.environ.get("OUT", "/tmp/rate")
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv, glob
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f))
         if "pcg" in r["Kernel_Name"] or "btd_direct" in r["Kernel_Name"]]
    d = [x[1] for x in sorted(d)]
    rec = json.load(open(os.path.join(OUT, "iters.json")))
    i = 0
    for B, its in rec:
        its = np.array(its, float)
        y = np.array(d[i:i + len(its)])
        i += len(its)
        its, y = its[3:], y[3:]
        b, a = np.polyfit(its, y, 1)
        print("B=%4d: %d launches, max-iteration counts %d..%d: duration = %.1f us + %.3f us x iterations (rms residual %.1f us)" % (
            B, len(its), its.min(), its.max(), a, b, np.sqrt(np.mean((a + b * its - y) ** 2))))
    sys.exit(0)
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
plant = os.environ.get("PLANT", "indy7")
N = int(os.environ.get("KNOTS", "32"))
rec = []
BATCHES = [tuple(int(v) for v in t.split(":")) for t in os.environ.get("BATCHES", "1:24,16:12,1024:3").split(",")]
for B, reps in BATCHES:
    its = []
    s = NativeSolver(plant, N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10))
    if os.environ.get("DIRECT"):
        s.set_linear_solver("direct")
    for r in range(reps):
        pr = fig8_problem(plant, N, B, seed=r)
        s.reset_dual(); s.reset_rho()
        out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
        its += [int(v) for v in out["pcg_iters_all"].max(axis=1)]
    rec.append((B, its))
os.makedirs(OUT, exist_ok=True)
json.dump(rec, open(os.path.join(OUT, "iters.json"), "w"))
