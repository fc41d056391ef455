#!/usr/bin/env python3
"""How much of a PCG launch is its tail?  For every PCG launch of a default 10-iteration solve: the iterations the trajectories need (their sum = the work),
the launch's longest trajectory (its duration), and how many trajectories are still iterating after 25 / 50 / 75 % of it -- C2, C3, C5 (GPU box).
    python tools/tail_accounting.py > profiles/r05_tail_accounting.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem, hparam_problem

CFG = {"C2 indy7 N=32 B=1024": ("indy7", 32, 1024, None), "C3 iiwa14 N=128 B=256": ("iiwa14", 128, 256, None), "C5 iiwa14 N=64 B=512 (shard 0 of the sweep)": ("iiwa14", 64, 512, 0)}
for name, (plant, N, B, shard) in CFG.items():
    if shard is None:
        pr, p, dt = fig8_problem(plant, N, B), dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10), 0.01
    else:
        pr = hparam_problem(plant, N, B, shard=shard); p, dt = dict(pr["params"]), pr["dt"]
    s = NativeSolver(plant, N, B, dt=dt, **p)
    if "rho" in pr: s.set_rho_penalty_batch(pr["rho"])
    s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    s.set_profiling(True); s.reset_dual(); s.reset_rho()
    r = s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    st = s.stage_times_us()
    it = r["pcg_iters_all"].astype(np.int64)          # [launch][trajectory]
    print("%s: solve %.0f us (stage clock), of which the 10 PCG launches %.0f us" % (name, st["total"], st["pcg"]))
    print("  launch   mean  median    p90    p99    max   work / (max x B)   still iterating after 25 / 50 / 75 %% of the longest")
    for i, row in enumerate(it):
        mx = row.max()
        alive = [int((row > f * mx).sum()) for f in (0.25, 0.5, 0.75)]
        print("  %6d %6.1f %7.0f %6.0f %6.0f %6d   %15.2f   %6d %6d %6d" % (i, row.mean(), np.median(row), np.quantile(row, .9), np.quantile(row, .99), mx, row.sum() / (mx * len(row)), *alive))
    print("  whole solve: sum of the launches' longest = %d iterations, the mean trajectory needs %.0f, the hardest single trajectory %d;" % (it.max(axis=1).sum(), it.sum(axis=0).mean(), it.sum(axis=0).max()))
    print("  trajectory-iterations done / (longest x B) summed over the launches = %.2f: the share of the PCG launches' slot-time in which a trajectory's slot still works\n" % (it.sum() / (it.max(axis=1).sum() * it.shape[1])))
