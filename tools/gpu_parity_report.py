#!/usr/bin/env python3
"""Measured parity numbers for DESIGN.md section 3 (runs on the MI355X box): the HIP path against the fp32 oracle and against the
float64 build of the oracle, next to the oracle's own fp32-vs-float64 gap, after 1/2/3/10 SQP iterations with PCG at its floor and
at the default tolerance.  Writes gpurun_out/parity_report.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd._lib import NativeSolver  # noqa: E402
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402


def traj_err(a, b):
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    return np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))


def merr(a, b):
    return np.abs(np.asarray(a, np.float64) - b) / np.maximum(1.0, np.abs(b))


rows = []
for plant, N, B in (("indy7", 32, 16), ("iiwa14", 32, 8), ("iiwa14", 64, 4), ("iiwa14", 128, 4)):
    pr = fig8_problem(plant, N, B)
    for tight in (True, False):
        for iters in (1, 2, 3, 10):
            p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters)
            if tight:
                p.update(pcg_tol=1e-9, max_pcg_iters=1000)
            g = NativeSolver(plant, N, B, dt=0.01, **p).solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
            o32 = OracleSolver(plant, N, B, dt=0.01, **p).solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
            o64 = OracleSolver(plant, N, B, dt=0.01, f64=True, **p).solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
            r = dict(plant=plant, N=N, B=B, iters=iters, pcg="floor" if tight else "1e-4",
                     gpu_vs_o32=float(traj_err(g["XU"], o32["XU"]).max()), gpu_vs_o64=float(traj_err(g["XU"], o64["XU"]).max()),
                     o32_vs_o64=float(traj_err(o32["XU"], o64["XU"]).max()),
                     merit_gpu_vs_o32=float(merr(g["final_merit"], o32["final_merit"]).max()),
                     merit_gpu_vs_o64=float(merr(g["final_merit"], o64["final_merit"]).max()),
                     merit_o32_vs_o64=float(merr(o32["final_merit"], o64["final_merit"]).max()),
                     steps_gpu_eq_o32=int(np.all(g["ls_step_size"] == o32["ls_step_size"], axis=0).sum()),
                     steps_o32_eq_o64=int(np.all(o32["ls_step_size"] == o64["ls_step_size"], axis=0).sum()),
                     pcg_diff_first=int(np.abs(g["pcg_iters"][0].astype(int) - o32["pcg_iters"][0]).max()))
            rows.append(r)
            print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w"), indent=1)
