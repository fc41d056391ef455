#!/usr/bin/env python3
"""Quick device timing of the full solve with per-stage hipEvent breakdown (debug aid; needs a GPU)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver  # noqa: E402
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--plant", default="indy7")
ap.add_argument("-N", type=int, default=32)
ap.add_argument("-B", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--max-pcg", type=int, default=-1)
ap.add_argument("--c5", action="store_true", help="BASELINE config C5's settings (hyper-parameter sweep shard: per-trajectory rho, dt 0.05, mu 1, pcg_tol 1e-3)")
ap.add_argument("--solver", default="pcg", choices=["pcg", "direct"])
a = ap.parse_args()
p = dict(DEFAULT_SOLVER_PARAMS)
p["max_sqp_iters"] = a.iters
if a.max_pcg >= 0:
    p["max_pcg_iters"] = a.max_pcg
dt = 0.01
if a.c5:
    from gato_amd.bsqp.workloads import hparam_problem
    pr = hparam_problem(a.plant, a.N, a.B, shard=0)
    p = dict(pr["params"], max_sqp_iters=a.iters)
    dt = pr["dt"]
else:
    pr = fig8_problem(a.plant, a.N, a.B)
s = NativeSolver(a.plant, a.N, a.B, dt=dt, **p)
if a.c5:
    s.set_rho_penalty_batch(pr["rho"])
s.set_linear_solver(a.solver)
s.set_profiling(True)
ts = []
for r in range(a.reps):
    s.reset_dual(); s.reset_rho()
    out = s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    ts.append(out["sqp_time_us"])
    st = s.stage_times_us()
print(a.plant, "N", a.N, "B", a.B, "iters", out["iters_done"], "ls", out["ls_num_iters"], "mean pcg", out["pcg_iters_all"].mean())
print("sqp_time_us:", ["%.0f" % t for t in ts], " -> traj-iter/s %.3e" % (a.B * out["iters_done"] / (min(ts) * 1e-6)))
print("stage us:", {k: round(v, 1) for k, v in st.items()})
print("merit", out["initial_merit"][:3], "->", out["final_merit"][:3])
pi = out["pcg_iters_all"]
print("pcg iters per SQP iteration: mean", pi.mean(axis=1).round(1), "max", pi.max(axis=1))
