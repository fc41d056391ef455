#!/usr/bin/env python3
"""Experiment (round 4): the C2 batch as G interleaved sub-batches (trajectory b in group b % G: equal difficulty mix), each with its own handle on
its own stream -- does one group's assembly / step / launch boundaries hide in another group's PCG tail?"""
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
def dev_t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def timed(groups, reps=50, pair=None, host_threads=False):
    hs, bufs, streams = [], [], []
    for idx in groups:
        if pair is not None: os.environ["GATO_PCG_PAIR"] = pair
        s = NativeSolver("indy7", N, len(idx), dt=0.01, **p)
        os.environ.pop("GATO_PCG_PAIR", None)
        hs.append(s)
        bufs.append((dev_t(pr["xu"][idx]), torch.empty((len(idx), s.traj), device=dev), dev_t(pr["x_s"][idx]), dev_t(pr["ref"][idx])))
        streams.append(torch.cuda.Stream())
    def step():
        for s, (x0, x, xs, ref), st in zip(hs, bufs, streams):
            with torch.cuda.stream(st):
                s.reset_async(True, True, st.cuda_stream)
                x.copy_(x0)
                s.solve_device(x.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st.cuda_stream)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, [b[1].cpu().numpy() for b in bufs]
base, xo = timed([np.arange(B)])
print("one handle, one stream: %.1f us per solve" % (base * 1e6))
for G in (2, 4):
    for pair in ("0", None):
        groups = [np.arange(g, B, G) for g in range(G)]
        t, xs = timed(groups, pair=pair)
        same = all(np.array_equal(x, xo[0][idx]) for x, idx in zip(xs, groups))
        print("G = %d interleaved groups, pair form %s: %.1f us per solve of the whole batch (%+.1f %%), bits equal: %s" % (G, "off" if pair == "0" else "auto", t * 1e6, 100 * (t / base - 1), same))
# offset start: group 1 enqueued half a PCG launch later is what the streams do by themselves after the first iteration
