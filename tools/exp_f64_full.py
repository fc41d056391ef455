#!/usr/bin/env python3
"""fp32 HIP path against its own float64 build (libgato_hip_f64.so, equal to the float64 oracle: tests/test_f64_gpu.py) at the BASELINE
sizes, EVERY trajectory; next to it the fp32 ORACLE on the first 32 trajectories against the same float64 results.
    python tools/exp_f64_full.py [--out profiles/r02_parity_full_size.json]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from oracle.oracle import OracleSolver
ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--oracle-sample", type=int, default=32)
a_ = ap.parse_args()
rows = []


def terr(x, y):
    return np.abs(np.asarray(x, np.float64) - y).max(axis=1) / np.maximum(1.0, np.abs(y).max(axis=1))


for name, plant, N, B in (("C2", "indy7", 32, 1024), ("C3", "iiwa14", 128, 256), ("C5 shard", "iiwa14", 64, 512)):
    pr = fig8_problem(plant, N, B)
    for iters, tight in ((1, True), (3, True), (10, False)):
        p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters)
        if tight:
            p.update(pcg_tol=1e-9, max_pcg_iters=1000)
        out = {}
        for f64 in (False, True):
            out[f64] = NativeSolver(plant, N, B, f64=f64, dt=0.01, **p).solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
        a, b = out[False], out[True]
        same = np.all(a["ls_step_size"] == b["ls_step_size"].astype(np.float32), axis=0)
        e = terr(a["XU"], b["XU"])
        row = {"config": name, "plant": plant, "N": N, "B": B, "sqp_iters": iters, "pcg": "floor (1e-9, 1000)" if tight else "default (1e-4, 200)",
               "hip_steps_equal_float64": int(same.sum()), "hip_err_max": float(e[same].max()), "hip_err_p99": float(np.quantile(e[same], 0.99)),
               "hip_err_median": float(np.median(e[same]))}
        S = min(a_.oracle_sample, B)
        if S and (tight or iters == 10):
            o = OracleSolver(plant, N, S, dt=0.01, **p)
            ro = o.solve(pr["xu"][:S], 0.01, pr["x_s"][:S], pr["ref"][:S])
            so = np.all(ro["ls_step_size"] == b["ls_step_size"][:, :S].astype(np.float32), axis=0)
            eo = terr(ro["XU"], b["XU"][:S])
            row.update(oracle_sample=S, oracle_steps_equal_float64=int(so.sum()), oracle_err_max=float(eo[so].max()) if so.any() else None,
                       oracle_err_median=float(np.median(eo[so])) if so.any() else None,
                       hip_err_max_same_sample=float(e[:S][same[:S]].max()) if same[:S].any() else None,
                       hip_err_median_same_sample=float(np.median(e[:S][same[:S]])) if same[:S].any() else None)
        rows.append(row)
        print(json.dumps(row), flush=True)
if a_.out:
    json.dump({"_note": "per-trajectory error = max |XU - XU_float64| / max(1, max |XU_float64|); statistics over the trajectories whose line-search "
                        "steps equal the float64 ones; float64 = libgato_hip_f64.so (the HIP kernels in double)", "rows": rows}, open(a_.out, "w"), indent=1)
