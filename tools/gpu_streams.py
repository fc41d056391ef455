#!/usr/bin/env python3
"""Experiment: G independent sub-batch solvers on G HIP streams vs one solver (debug aid; needs a GPU)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem

B, N, dt = 1024, 32, 0.01
p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10)
dev = torch.device("cuda", 0)
for G in (1, 2, 4, 8, 16):
    per = B // G
    solvers, bufs, streams = [], [], []
    for g in range(G):
        pr = fig8_problem("indy7", N, per, batch_offset=g * per)
        s = NativeSolver("indy7", N, per, dt=dt, **p)
        xu0 = torch.from_numpy(pr["xu"]).to(dev); xu = torch.empty_like(xu0)
        xs = torch.from_numpy(pr["x_s"]).to(dev); ref = torch.from_numpy(pr["ref"]).to(dev)
        solvers.append(s); bufs.append((xu0, xu, xs, ref)); streams.append(torch.cuda.Stream())
    def step():
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                xu0, xu, xs, ref = bufs[g]
                st = streams[g].cuda_stream
                solvers[g].reset_async(True, True, st)
                xu.copy_(xu0)
                solvers[g].solve_device(xu.data_ptr(), dt, xs.data_ptr(), ref.data_ptr(), st)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / K
    print("G=%2d  %.3f ms/solve  %.3e traj-iter/s" % (G, t * 1e3, B * 10 / t))
