#!/usr/bin/env python3
"""Sensitivity of the SQP iterates to rounding: the fp32 oracle against the float64 build of the SAME source on the same inputs
(CPU only).  Quantifies what "parity within 1e-4 after k iterations" can mean for this problem: two correct fp32 implementations
with different summation orders cannot agree better than fp32 agrees with float64.

    python tools/sensitivity.py            # table for DESIGN.md section 3
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402


def traj_err(a, b):
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    return np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))


def run(plant, N, B, iters, tight, fstd=0.0):
    over = dict(max_sqp_iters=iters)
    if tight:
        over.update(pcg_tol=1e-9, max_pcg_iters=1000)
    p = dict(DEFAULT_SOLVER_PARAMS, **over)
    pr = fig8_problem(plant, N, B, f_ext_std=fstd)
    out = []
    for f64 in (False, True):
        s = OracleSolver(plant, N, B, dt=0.01, f64=f64, **p)
        s.set_f_ext_batch(pr["f_ext"])
        out.append(s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"]))
    a, b = out
    e = traj_err(a["XU"], b["XU"])
    m = np.abs(a["final_merit"] - b["final_merit"]) / np.maximum(1.0, np.abs(b["final_merit"]))
    same = np.all(a["ls_step_size"] == b["ls_step_size"], axis=0)
    return e, m, same


if __name__ == "__main__":
    print("%-8s %4s %3s %5s %6s | %10s %10s %10s | %10s | %s" % ("plant", "N", "B", "iters", "pcg", "XU max", "XU median", "XU max(same)", "merit max", "same steps"))
    for plant, N, B in (("indy7", 32, 16), ("iiwa14", 32, 8), ("iiwa14", 64, 4), ("iiwa14", 128, 4)):
        for tight in (True, False):
            for iters in (1, 2, 3, 10):
                e, m, same = run(plant, N, B, iters, tight)
                print("%-8s %4d %3d %5d %6s | %10.2e %10.2e %10.2e | %10.2e | %d/%d" % (
                    plant, N, B, iters, "floor" if tight else "1e-4", e.max(), np.median(e), e[same].max() if same.any() else float("nan"), m.max(),
                    same.sum(), B))
