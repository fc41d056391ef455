"""dump one iiwa14 solve (and S, Pinv, gamma after it) to an npz; run once per library build and compare bits"""
import sys, numpy as np
sys.path.insert(0, ".")
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from gato_amd._lib import NativeSolver
out = {}
for (plant, N, B) in (("iiwa14", 128, 256), ("iiwa14", 64, 37), ("indy7", 32, 1024), ("indy7", 4, 3), ("indy7", 8, 5), ("indy7", 128, 3), ("iiwa14", 16, 9), ("iiwa14", 32, 70), ("indy7", 64, 40)):
    pr = fig8_problem(plant, N, B, f_ext_std=2.0 if N == 8 else 0.0)
    s = NativeSolver(plant, N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=5))
    s.set_f_ext_batch(pr["f_ext"])
    r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    for k in ("XU", "final_merit", "pcg_iters_all", "ls_step_size"):
        out[f"{plant}_{N}_{B}_{k}"] = np.asarray(r[k])
    for name in ("S", "Pinv", "gamma", "lambda", "dz", "q", "r"):
        out[f"{plant}_{N}_{B}_{name}"] = s.read(name)
np.savez(sys.argv[1], **out)
