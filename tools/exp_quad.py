"""C3's PCG iteration in the QUAD form against the ROLE form (pcgq_kernel vs pcgs_kernel<.., fold = false>, kernels.hpp): the same bits, and the launch
time at FIXED iteration counts (pcg_tol < 0: the exit test never passes) -> microseconds per iteration and per prologue.  Both without the in-kernel
fold (GATO_PCG_FOLD=0: schur2_kernel leaves the complete P^-1), so the iteration is what differs.  Runs on the MI355X box -> profiles/r06_c3_quad.txt."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem


def make(plant, N, B, quad, **over):
    env = {"GATO_PCG_VARIANT": "7", "GATO_PCG_FOLD": "0", "GATO_PCGS_QUAD": "1" if quad else "0"}
    os.environ.update(env)
    s = NativeSolver(plant, N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, **over))
    for k in env:
        del os.environ[k]
    return s


def rate(plant, N, B, quad, K, iters=3):
    pr = fig8_problem(plant, N, B)
    s = make(plant, N, B, quad, max_sqp_iters=iters, max_pcg_iters=K)
    s.set_pcg_tol_batch(np.full(B, -1.0, np.float32))
    s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    s.set_profiling(True)
    ts = []
    for _ in range(3):
        s.reset_dual(); s.reset_rho()
        r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
        ts.append(s.stage_times_us()["pcg"] / iters)
    assert r["pcg_iters_all"].min() == K and r["pcg_iters_all"].max() == K
    return min(ts)


out = []
for plant, N, B in (("iiwa14", 128, 256), ("iiwa14", 64, 256), ("indy7", 128, 256)):
    pr = fig8_problem(plant, N, B)
    res = {}
    for quad in (0, 1):
        s = make(plant, N, B, quad, max_sqp_iters=3)
        r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
        res[quad] = (r["XU"].copy(), s.read("lambda").copy(), r["pcg_iters_all"].copy(), r["final_merit"].copy())
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    row = {"plant": plant, "N": N, "B": B, "bit_identical_after_3_sqp_iterations": bool(same), "mean_pcg_iters": float(res[0][2].mean()),
           "max_abs_dlambda": float(np.abs(res[0][1] - res[1][1]).max())}
    for quad in (0, 1):
        t = {K: rate(plant, N, B, quad, K) for K in (10, 50, 90)}
        slope = (t[90] - t[10]) / 80.0
        row["quad" if quad else "role"] = {"launch_us": t, "us_per_iteration": slope, "prologue_us": t[10] - 10 * slope}
    row["iteration_quad_over_role"] = row["quad"]["us_per_iteration"] / row["role"]["us_per_iteration"]
    out.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_c3_quad.json"), "w"), indent=1)
