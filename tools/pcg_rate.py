"""PCG launch time against a FIXED iteration count for every trajectory (pcg_tol < 0: the exit test never passes), per form and residency."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem, hparam_problem

def run(plant, N, B, K, env, iters=3, hparam=False):
    for k, v in env.items(): os.environ[k] = v
    if hparam:
        pr = hparam_problem(plant, N, B, shard=3); p = dict(pr["params"]); dt = pr["dt"]
    else:
        pr = fig8_problem(plant, N, B); p = dict(DEFAULT_SOLVER_PARAMS); dt = 0.01
    p.update(max_sqp_iters=iters, max_pcg_iters=K)
    s = NativeSolver(plant, N, B, dt=dt, **p)
    for k in env: del os.environ[k]
    if "rho" in pr: s.set_rho_penalty_batch(pr["rho"])
    s.set_pcg_tol_batch(np.full(B, -1.0, np.float32))
    s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    s.set_profiling(True)
    ts = []
    for _ in range(3):
        s.reset_dual(); s.reset_rho()
        r = s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
        ts.append(s.stage_times_us()["pcg"] / iters)
    assert r["pcg_iters_all"].min() == K and r["pcg_iters_all"].max() == K, (r["pcg_iters_all"].min(), r["pcg_iters_all"].max())
    return min(ts)

cfgs = [("indy7", 32, 1024, {"GATO_PCG_PAIR": "0"}, "single-lane, 2 wavefronts / SIMD (C2 as launched)"),
        ("indy7", 32, 512, {"GATO_PCG_PAIR": "0"}, "single-lane, 1 wavefront / SIMD"),
        ("indy7", 32, 256, {"GATO_PCG_PAIR": "0"}, "single-lane, 1 wavefront on every other SIMD"),
        ("indy7", 32, 512, {"GATO_PCG_PAIR": "1"}, "pair form, 2 wavefronts / SIMD (all resident at 246 registers)"),
        ("indy7", 32, 256, {"GATO_PCG_PAIR": "1"}, "pair form, 1 wavefront / SIMD"),
        ("indy7", 32, 64, {"GATO_PCG_PAIR": "1"}, "pair form, a quarter of the CUs"),
        ("indy7", 32, 64, {"GATO_PCG_PAIR": "0"}, "single-lane, a quarter of the CUs")]
out = []
for plant, N, B, env, what in cfgs:
    t = {K: run(plant, N, B, K, env) for K in (10, 50, 90)}
    slope = (t[90] - t[10]) / 80.0
    out.append(dict(plant=plant, N=N, B=B, env=env, what=what, launch_us=t, us_per_iteration=slope, prologue_us=t[10] - 10 * slope))
    print("%-70s B=%4d  launch us %s   %.3f us / iteration, prologue %.1f us" % (what, B, {k: round(v, 1) for k, v in t.items()}, slope, t[10] - 10 * slope), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_pcg_rate.json"), "w"), indent=1)
