"""PCG iteration counts and the stage clock of one-iteration indy7 N = 128 solves at B = 1 / 8 from the reset state (where the N = 128, B <= 8 cells of the
MPC heat-map spend their time: DESIGN.md section 4).  Runs on the MI355X box -> profiles/r06_n128_small_batch.txt."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
for B in (1, 8):
    pr = fig8_problem("indy7", 128, B)
    s = NativeSolver("indy7", 128, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1))
    s.set_profiling(True)
    for rep in range(5):
        s.reset_dual(); s.reset_rho()
        r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    print("indy7 N=128 B=%d one-iteration solve from the reset state: PCG iterations %s, stage us %s, sqp_time_us %.1f" % (B, r["pcg_iters_all"][0].tolist(), {k: round(v, 1) for k, v in s.stage_times_us().items()}, r["sqp_time_us"]))
