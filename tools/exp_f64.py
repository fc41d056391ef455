#!/usr/bin/env python3
"""Float64 build of the HIP path (libgato_hip_f64.so) against the float64 oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from oracle.oracle import OracleSolver
for plant, N, B, iters in (("indy7", 8, 2, 1), ("indy7", 32, 4, 3), ("iiwa14", 16, 3, 3), ("indy7", 64, 2, 2), ("iiwa14", 128, 2, 2)):
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters, pcg_tol=1e-9, max_pcg_iters=1000)
    pr = fig8_problem(plant, N, B, f_ext_std=2.0)
    t0 = time.time()
    nat = NativeSolver(plant, N, B, f64=True, dt=0.01, **p)
    o64 = OracleSolver(plant, N, B, dt=0.01, f64=True, **p)
    for s in (nat, o64):
        s.set_f_ext_batch(pr["f_ext"])
    rg = nat.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    ro = o64.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    e = np.abs(rg["XU"] - ro["XU"]).max() / np.abs(ro["XU"]).max()
    print("%s N=%d B=%d %d iterations: XU rel err vs float64 oracle %.2e, steps equal %s, pcg iters %s vs %s, merit rel err %.2e, %.1f s, solve %.0f us" % (
        plant, N, B, iters, e, np.array_equal(rg["ls_step_size"], ro["ls_step_size"]), rg["pcg_iters"].max(axis=1), ro["pcg_iters"].max(axis=1),
        np.abs(rg["final_merit"] - ro["final_merit"]).max() / np.abs(ro["final_merit"]).max(), time.time() - t0, rg["sqp_time_us"]), flush=True)
