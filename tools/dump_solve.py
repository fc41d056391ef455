"""dump one iiwa14 solve (and S, Pinv, gamma after it) to an npz; run once per library build and compare bits"""
import sys, numpy as np
sys.path.insert(0, ".")
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from gato_amd._lib import NativeSolver
out = {}
for (N, B) in ((128, 256), (64, 37)):
    pr = fig8_problem("iiwa14", N, B)
    s = NativeSolver("iiwa14", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=5))
    r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    for k in ("XU", "final_merit", "pcg_iters_all", "ls_step_size"):
        out[f"{N}_{B}_{k}"] = np.asarray(r[k])
    for name in ("S", "Pinv", "gamma", "lambda"):
        out[f"{N}_{B}_{name}"] = s.read(name)
np.savez(sys.argv[1], **out)
