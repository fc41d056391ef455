#!/usr/bin/env python3
"""fp32 HIP path against its own float64 build at the BASELINE sizes (all trajectories): one iteration with PCG at its floor, and 10 default iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
for plant, N, B in (("indy7", 32, 1024), ("iiwa14", 128, 256), ("iiwa14", 64, 512)):
    pr = fig8_problem(plant, N, B)
    for iters, tight in ((1, True), (3, True), (10, False)):
        p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters)
        if tight:
            p.update(pcg_tol=1e-9, max_pcg_iters=1000)
        out = {}
        for f64 in (False, True):
            s = NativeSolver(plant, N, B, f64=f64, dt=0.01, **p)
            t0 = time.time()
            out[f64] = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
            out[f64]["wall"] = time.time() - t0
        a, b = out[False], out[True]
        e = np.abs(a["XU"].astype(np.float64) - b["XU"]).max(axis=1) / np.maximum(1.0, np.abs(b["XU"]).max(axis=1))
        same = np.all(a["ls_step_size"] == b["ls_step_size"].astype(np.float32), axis=0)
        print("%s N=%d B=%d, %d iteration(s), %s: steps equal on %d/%d trajectories; on those XU error max %.2e, 99%% %.2e, median %.2e; "
              "fp64 solve %.1f ms (fp32 %.2f ms)" % (plant, N, B, iters, "PCG at its floor" if tight else "default tolerances", same.sum(), B,
                                                      e[same].max(), np.quantile(e[same], 0.99), np.median(e[same]), b["sqp_time_us"] / 1e3, a["sqp_time_us"] / 1e3), flush=True)
