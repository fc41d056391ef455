#!/usr/bin/env python3
"""Teacher-forced trace (runs on the MI355X box): ten SQP iterations, each started on the HIP path, the fp32 oracle and the float64 oracle
from the fp32 oracle's state; prints per iteration the PCG counts, the dz errors of the three pairs and the steps.  Produced
profiles/r02_teacher_forced_indy7_{floor,default}.log:  python tools/teacher_forced_trace.py indy7 32 8 1 | 0"""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from oracle.oracle import OracleSolver
plant, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tight = int(sys.argv[4])
DT = 0.01
p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1)
if tight: p.update(pcg_tol=1e-9, max_pcg_iters=1000)
pr = fig8_problem(plant, N, B, f_ext_std=2.0)
nat = NativeSolver(plant, N, B, dt=DT, **p); orc = OracleSolver(plant, N, B, dt=DT, **p); o64 = OracleSolver(plant, N, B, dt=DT, f64=True, **p)
for s in (nat, orc, o64): s.set_f_ext_batch(pr["f_ext"])
xs, ref = pr["x_s"], pr["ref"]; xu = pr["xu"].copy()
lam = np.zeros((B, N + 2, nat.nx), np.float32); rho = np.full(B, 0.01, np.float32); drho = np.ones(B, np.float32)
f = np.float32
for it in range(10):
    for s in (nat, orc, o64):
        s.set_rho_penalty_batch(rho, False); s.set_drho_batch(drho, False)
    nat.write("lambda", lam); orc.set_lambda(lam); o64.set_lambda(lam)
    rg = nat.solve(xu, DT, xs, ref); ro = orc.solve(xu, DT, xs, ref); r6 = o64.solve(xu, DT, xs, ref)
    dg, do, d6 = nat.read("dz").reshape(B, -1), orc.buf("dz"), o64.buf("dz")
    sc = np.maximum(1e-3, np.abs(d6).max(axis=1))
    print("it", it, "rho", rho[:4], "pcg g", rg["pcg_iters"][0], "o32", ro["pcg_iters"][0], "o64", r6["pcg_iters"][0])
    print("   dz err g-o32", (np.abs(dg - do).max(axis=1) / sc).round(5), "\n   g-o64", (np.abs(dg - d6).max(axis=1) / sc).round(5), "\n   o32-o64", (np.abs(do - d6).max(axis=1) / sc).round(5))
    print("   steps g", rg["ls_step_size"][0], "o32", ro["ls_step_size"][0], "o64", r6["ls_step_size"][0])
    success = ro["ls_step_size"][0] > 0
    mult = np.where(success, np.minimum(drho / f(1.2), f(1) / f(1.2)), np.maximum(drho * f(1.2), f(1.2))).astype(f)
    drho = mult; rho = np.minimum(np.maximum(rho * mult, f(1e-8)), f(10.0)).astype(f)
    xu, lam = ro["XU"].copy(), orc.buf("lambda")
