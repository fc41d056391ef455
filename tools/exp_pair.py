#!/usr/bin/env python3
"""Experiment: pair form of the fused PCG kernel (GATO_PCG_PAIR=1) against the single-lane form at C2: time per solve, PCG counts, iterates."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    N, B = 32, int(os.environ.get("BATCH", "1024"))
    pr = fig8_problem("indy7", N, B)
    s = NativeSolver("indy7", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10))
    out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    dev = torch.device("cuda", 0)
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0, x, xs, ref = t_(pr["xu"]), torch.empty((B, s.traj), device=dev), t_(pr["x_s"]), t_(pr["ref"])
    st = torch.cuda.current_stream()
    def step():
        s.reset_async(True, True, st.cuda_stream)
        x.copy_(x0)
        s.solve_device(x.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    np.savez(sys.argv[1], xu=out["XU"], it=out["pcg_iters_all"], merit=out["final_merit"], t=(time.perf_counter() - t0) / 20 * 1e6)
else:
    for v in ("0", "1"):
        subprocess.check_call([sys.executable, __file__, "/tmp/pair%s.npz" % v], env=dict(os.environ, GATO_PCG_PAIR=v))
    a, b = np.load("/tmp/pair0.npz"), np.load("/tmp/pair1.npz")
    d = np.abs(a["xu"] - b["xu"]).max(axis=1)
    print("single-lane form %.1f us per solve, pair form %.1f us per solve" % (a["t"], b["t"]))
    print("PCG counts: max |diff| %d, differing launches %d of %d; per-iteration max %s vs %s" % (
        np.abs(a["it"] - b["it"]).max(), (a["it"] != b["it"]).sum(), a["it"].size, a["it"].max(axis=1), b["it"].max(axis=1)))
    print("iterates after 10 iterations: median max-abs diff %.2e, 90%% %.2e; final merit median rel diff %.2e" % (
        np.median(d), np.quantile(d, 0.9), np.median(np.abs(a["merit"] - b["merit"]) / np.abs(a["merit"]))))
