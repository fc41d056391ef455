#!/usr/bin/env python3
"""Latency of small batches: ms per solve and per SQP iteration for B in {1..64} (device-synchronised host wall clock, best of reps)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
ap = argparse.ArgumentParser()
ap.add_argument("-N", type=int, default=32)
ap.add_argument("--iters", type=int, nargs="+", default=[1, 5])
ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32, 64])
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--graph", type=int, default=0)
a = ap.parse_args()
for it in a.iters:
    for B in a.batches:
        p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=it)
        pr = fig8_problem("indy7", a.N, B)
        s = NativeSolver("indy7", a.N, B, dt=0.01, **p)
        s.set_graph_mode(a.graph)
        ts = []
        for r in range(a.reps):
            s.reset_dual(); s.reset_rho()
            ts.append(s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])["sqp_time_us"])
        ts = np.array(ts[3:])
        print("N %d B %3d iters %d: solve min %.1f us median %.1f us  (%.1f us / SQP iteration)" % (a.N, B, it, ts.min(), np.median(ts), np.median(ts) / it), flush=True)
