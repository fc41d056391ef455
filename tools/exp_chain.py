#!/usr/bin/env python3
"""Experiment: the latency chain of the hardest trajectories of the C2 batch, solved alone (B = 1, 8, 32): wall time per solve, PCG
iterations, per-stage times (profiling mode) -- what bounds the full batch's solve."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
dev = torch.device("cuda", 0)
N, B = 32, 1024
p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10)
pr = fig8_problem("indy7", N, B)
full = NativeSolver("indy7", N, B, dt=0.01, **p)
out = full.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
tot = out["pcg_iters_all"].sum(axis=0)
order = np.argsort(-tot)


def dev_t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


for H in (1, 8, 32, 128, 1024):
    idx = np.sort(order[:H])
    s = NativeSolver("indy7", N, H, dt=0.01, **p)
    x0, x, xs, ref = dev_t(pr["xu"][idx]), torch.empty((H, full.traj), device=dev), dev_t(pr["x_s"][idx]), dev_t(pr["ref"][idx])
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
    t = (time.perf_counter() - t0) / 20 * 1e6
    its = out["pcg_iters_all"][:, idx]
    s.set_profiling(True)
    s.reset_dual(); s.reset_rho()
    s.solve(pr["xu"][idx], 0.01, pr["x_s"][idx], pr["ref"][idx])
    stg = s.stage_times_us()
    print("hardest %4d: %.1f us per solve; sum over iterations of the max PCG count %d (one trajectory's max total %d); (t - 0.75*sum)/10 = %.1f us; "
          "stages (profiling mode, events around every launch) %s" % (H, t, its.max(axis=1).sum(), its.sum(axis=0).max(), (t - 0.75 * its.max(axis=1).sum()) / 10,
                                                                    {k: round(v, 1) for k, v in stg.items()}), flush=True)
