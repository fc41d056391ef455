#!/usr/bin/env python3
"""Experiment: how much does decoupling hard from easy trajectories buy at C2?  The batch is split by total PCG iterations (known
from a first solve) into a hard and an easy sub-batch, solved by two handles on two streams concurrently."""
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
print("total pcg iters: max %d mean %.1f; sum of per-launch max %d" % (tot.max(), tot.mean(), out["pcg_iters_all"].max(axis=1).sum()))

def dev_t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

import ctypes
_hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(cu_lo, cu_hi, total=256):
    """a stream whose kernels only run on CUs [cu_lo, cu_hi) (hipExtStreamCreateWithCUMask)"""
    words = (total + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for c in range(cu_lo, cu_hi):
        m[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def timed(groups, prio, reps=20, masks=None):
    hs, bufs, streams = [], [], []
    for gi, idx in enumerate(groups):
        s = NativeSolver("indy7", N, len(idx), dt=0.01, **p)
        hs.append(s)
        bufs.append((dev_t(pr["xu"][idx]), torch.empty((len(idx), full.traj), device=dev), dev_t(pr["x_s"][idx]), dev_t(pr["ref"][idx])))
        streams.append(masked_stream(*masks[gi]) if masks else torch.cuda.Stream(priority=prio[gi]))
    def step():
        for s, (x0, x, xs, ref), st in zip(hs, bufs, streams):
            with torch.cuda.stream(st):
                s.reset_async(True, True, st.cuda_stream)
                x.copy_(x0)
                s.solve_device(x.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    xs_out = [b[1].cpu().numpy() for b in bufs]
    return dt, xs_out

t_full, x_full = timed([np.arange(B)], [0])
print("single batch: %.1f us per solve" % (t_full * 1e6))
for H in (16, 32, 64, 128, 256, 512):
    hard, easy = np.sort(order[:H]), np.sort(order[H:])
    for prio in ((0, 0), (0, -1)):
        t, xo = timed([easy, hard], prio)
        ok = np.array_equal(xo[0], x_full[0][easy]) and np.array_equal(xo[1], x_full[0][hard])
        print("H=%4d prio %s: %.1f us per solve (%.2fx)  bit-equal to the single batch: %s" % (H, prio, t * 1e6, t_full / t, ok))

print("CU-masked streams (hipExtStreamCreateWithCUMask): hard sub-batch on its own CUs")
t_m, _ = timed([np.arange(B)], [0], masks=[(0, 256)])
print("single batch on a stream masked to all 256 CUs: %.1f us" % (t_m * 1e6))
for stride in (False, True):
  for H, C in ((32, 32), (64, 32), (64, 64), (128, 64), (128, 32), (256, 64), (256, 128)):
    hard, easy = np.sort(order[:H]), np.sort(order[H:])
    t, xo = timed([easy, hard], (0, 0), masks=[(C, 256), (0, C)])
    ok = np.array_equal(xo[0], x_full[0][easy]) and np.array_equal(xo[1], x_full[0][hard])
    print("H=%4d on %3d CUs, rest on %3d: %.1f us per solve (%.2fx)  bit-equal: %s" % (H, C, 256 - C, t * 1e6, t_full / t, ok), flush=True)
  break
# the hard sub-batch alone on its CU partition: its own chain
for H, C in ((32, 32), (64, 32), (64, 64), (128, 64)):
    hard = np.sort(order[:H])
    t, _ = timed([hard], (0,), masks=[(0, C)])
    print("hard %d alone on %d CUs: %.1f us" % (H, C, t * 1e6), flush=True)
for H, C in ((32, 32), (64, 32), (64, 64), (128, 64)):
    easy = np.sort(order[H:])
    t, _ = timed([easy], (0,), masks=[(C, 256)])
    print("easy %d alone on %d CUs: %.1f us" % (B - H, 256 - C, t * 1e6), flush=True)
