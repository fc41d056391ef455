#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path against the CPU oracle on one problem (debug aid; needs a GPU)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver  # noqa: E402
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plant", default="indy7")
    ap.add_argument("-N", type=int, default=8)
    ap.add_argument("-B", type=int, default=4)
    ap.add_argument("--fext", type=float, default=0.0)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    p = dict(DEFAULT_SOLVER_PARAMS)
    p["max_sqp_iters"] = a.iters
    dt = 0.01
    pr = fig8_problem(a.plant, a.N, a.B, f_ext_std=a.fext)
    nat = NativeSolver(a.plant, a.N, a.B, dt=dt, **p)
    orc = OracleSolver(a.plant, a.N, a.B, dt=dt, **p)
    nat.set_f_ext_batch(pr["f_ext"])
    orc.set_f_ext_batch(pr["f_ext"])
    xu, xs, ref = pr["xu"], pr["x_s"], pr["ref"]
    B, N, nx, nu, nq = a.B, a.N, nat.nx, nat.nu, nat.nq

    # merit(1)
    nat.stage("merit1", xu, dt, xs, ref)
    m_o = orc.merit(xu, xs, ref, dt, num_alphas=1, zero_dz=True)[:, 0]
    print("merit1      ", relmax(nat.read("merit_cur"), m_o))
    # kkt
    nat.stage("kkt", xu, dt, xs, ref)
    orc.setup_kkt(xu, xs, ref, dt)
    dk = nat.dense_kkt(dt)
    for name in ("A", "B", "c", "Q", "q", "R", "r"):
        o = orc.buf(name)
        g = dk[name]
        if name in ("A", "B"):
            o = o[:, :N - 1]
            g = g[:, :N - 1]
        if name in ("R", "r"):
            o = o[:, :N - 1]
            g = g[:, :N - 1]
        print("kkt %-8s" % name, relmax(g, o))
    # schur (same KKT inputs on both sides: the oracle's own vs the GPU's own, both from identical xu)
    nat.stage("schur", xu, dt, xs, ref)
    orc.form_schur()
    dk = nat.dense_kkt(dt)
    print("schur Qinv  ", relmax(dk["Qinv"], orc.buf("Qinv")))
    print("schur Rinv  ", relmax(dk["Rinv"][:, :N - 1], orc.buf("Rinv")[:, :N - 1]))
    for name in ("S", "Pinv", "gamma"):
        g = nat.read(name).reshape(orc.buf(name).shape)
        print("schur %-6s" % name, relmax(g, orc.buf(name)))
    # pcg
    nat.stage("pcg", xu, dt, xs, ref)
    orc.pcg()
    lam_g = nat.read("lambda").reshape(B, N + 2, nx)
    print("pcg lambda  ", relmax(lam_g, orc.buf("lambda")), "iters gpu", nat.read("pcg_iters"), "orc", orc.ibuf("pcg_iters", (B,)))
    # dz from the ORACLE's lambda on both sides
    nat.write("lambda", orc.buf("lambda"))
    nat.stage("dz", xu, dt, xs, ref)
    orc.compute_dz()
    print("dz          ", relmax(nat.read("dz").reshape(B, -1), orc.buf("dz")))
    print("resid q     ", relmax(nat.read("q").reshape(B, N, nx), orc.buf("q")))
    # merit(8) from the oracle's dz
    nat.write("dz", orc.buf("dz"))
    nat.stage("merit8", xu, dt, xs, ref)
    m8 = orc.merit(xu, xs, ref, dt, num_alphas=8)
    print("merit8      ", relmax(nat.read("merit").reshape(B, 8), m8))
    # full solve
    nat.reset_dual(); nat.reset_rho(); orc.reset_dual(); orc.reset_rho()
    t = time.time()
    rg = nat.solve(xu, dt, xs, ref)
    tg = time.time() - t
    ro = orc.solve(xu, dt, xs, ref)
    print("solve: gpu %.1f us (%.3f s wall)  oracle %.1f us" % (rg["sqp_time_us"], tg, ro["sqp_time_us"]))
    print("  XU rel     ", rel(rg["XU"], ro["XU"]))
    print("  final merit", rg["final_merit"][:4], ro["final_merit"][:4])
    print("  steps gpu\n", rg["ls_step_size"].T[:4], "\n  steps orc\n", ro["ls_step_size"].T[:4])
    print("  pcg gpu\n", rg["pcg_iters_all"].T[:4], "\n  pcg orc\n", ro["pcg_iters_all"].T[:4])


if __name__ == "__main__":
    main()
