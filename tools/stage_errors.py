#!/usr/bin/env python3
"""Where do iiwa14's 1.4-1.6e-4 (one SQP iteration, PCG at its floor, HIP vs fp32 oracle) come from?  (round-3 review, item 5)

Three implementations of the same iteration -- the HIP kernels (fp32), the CPU oracle (fp32), the CPU oracle (float64, the arbiter) -- each
running its OWN pipeline from identical inputs; after every stage the stage's tensor is compared with the float64 one:
    per trajectory  max|a - a64| / max|a64|      -> median and max over the batch
so the table shows at which stage the error ENTERS and how the two fp32 paths compare there.  Two derived rows separate conditioning from
arithmetic:  `lambda | exact solve of own (S, gamma)` is the float64 dense solve of each path's own fp32 Schur system (what a perfect linear
solver would return for the perturbed system) and `lambda | own PCG` what its PCG returned.

    python tools/stage_errors.py            (on the GPU box; writes gpurun_out/r04_stage_errors.{json,txt})
"""
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

DT = 0.01


def terr(a, b):
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    return np.abs(a - b).max(axis=1) / np.maximum(1e-300, np.abs(b).max(axis=1))


def dense_S(Srows, N, nx):
    """[N][nx][3nx] block rows (reference layout) -> dense (N nx)^2"""
    M = np.zeros((N * nx, N * nx))
    for k in range(N):
        r = slice(k * nx, (k + 1) * nx)
        if k > 0:
            M[r, (k - 1) * nx:k * nx] = Srows[k][:, :nx]
        M[r, k * nx:(k + 1) * nx] = Srows[k][:, nx:2 * nx]
        if k < N - 1:
            M[r, (k + 1) * nx:(k + 2) * nx] = Srows[k][:, 2 * nx:]
    return M


def run(plant, N, B):
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1, pcg_tol=1e-9, max_pcg_iters=1000)
    pr = fig8_problem(plant, N, B)
    xu, xs, ref = pr["xu"], pr["x_s"], pr["ref"]
    nat = NativeSolver(plant, N, B, dt=DT, **p)
    o32 = OracleSolver(plant, N, B, dt=DT, **p)
    o64 = OracleSolver(plant, N, B, dt=DT, f64=True, **p)
    nx = nat.nx
    st = {"hip": {}, "o32": {}, "o64": {}}
    # stage by stage, every path on its own upstream
    nat.stage("kkt", xu, DT, xs, ref)
    nat.stage("schur", xu, DT, xs, ref)
    dk = nat.dense_kkt(DT)
    for o, key in ((o32, "o32"), (o64, "o64")):
        o.setup_kkt(xu, xs, ref, DT)
        for n in ("A", "B", "c", "Q", "q", "R", "r"):
            st[key][n] = o.buf(n)
        o.form_schur()
        for n in ("Qinv", "Rinv", "S", "Pinv", "gamma"):
            st[key][n] = o.buf(n)
    for n in ("A", "B", "c", "Q", "q", "R", "r", "Qinv", "Rinv"):
        st["hip"][n] = dk[n]
    for n in ("S", "Pinv", "gamma"):
        st["hip"][n] = nat.read(n).reshape(st["o64"][n].shape)
    nat.stage("pcg", xu, DT, xs, ref)
    st["hip"]["lambda"] = nat.read("lambda").reshape(B, N + 2, nx)
    st["hip"]["pcg_iters"] = nat.read("pcg_iters")
    nat.stage("dz", xu, DT, xs, ref)
    st["hip"]["dz"] = nat.read("dz").reshape(B, -1)
    for o, key in ((o32, "o32"), (o64, "o64")):
        o.pcg()
        st[key]["lambda"] = o.buf("lambda")
        st[key]["pcg_iters"] = o.ibuf("pcg_iters", (B,))
        o.compute_dz()
        st[key]["dz"] = o.buf("dz")
    # whole iteration (fresh solvers: the stage calls above changed lambda)
    full = {}
    for key, s in (("hip", NativeSolver(plant, N, B, dt=DT, **p)), ("o32", OracleSolver(plant, N, B, dt=DT, **p)), ("o64", OracleSolver(plant, N, B, dt=DT, f64=True, **p))):
        full[key] = s.solve(xu, DT, xs, ref)
    rows = []

    def add(name, fn):
        e = {k: fn(k) for k in ("hip", "o32")}
        rows.append({"stage": name, **{k + "_median": float(np.median(v)) for k, v in e.items()}, **{k + "_max": float(np.max(v)) for k, v in e.items()}})

    sl = {"A": slice(0, N - 1), "B": slice(0, N - 1), "R": slice(0, N - 1), "r": slice(0, N - 1), "Rinv": slice(0, N - 1)}
    for n in ("A", "B", "c", "Q", "q", "Qinv", "Rinv"):
        add(n, lambda k, n=n: terr(st[k][n][:, sl.get(n, slice(None))], st["o64"][n][:, sl.get(n, slice(None))]))
    add("S left (phi)", lambda k: terr(st[k]["S"][:, 1:, :, :nx], st["o64"]["S"][:, 1:, :, :nx]))
    add("S main (-theta)", lambda k: terr(st[k]["S"][:, :, :, nx:2 * nx], st["o64"]["S"][:, :, :, nx:2 * nx]))
    add("Pinv main ((theta+rho I)^-1)", lambda k: terr(st[k]["Pinv"][:, :, :, nx:2 * nx], st["o64"]["Pinv"][:, :, :, nx:2 * nx]))
    add("Pinv left (stair)", lambda k: terr(st[k]["Pinv"][:, 1:, :, :nx], st["o64"]["Pinv"][:, 1:, :, :nx]))
    add("gamma", lambda k: terr(st[k]["gamma"], st["o64"]["gamma"]))
    # conditioning vs arithmetic: the float64 solution of each path's OWN fp32 system
    exact = {}
    for k in ("hip", "o32", "o64"):
        lam = np.zeros((B, N * nx))
        for b in range(B):
            lam[b] = np.linalg.solve(dense_S(np.asarray(st[k]["S"][b], np.float64), N, nx), np.asarray(st[k]["gamma"][b, 1:N + 1], np.float64).reshape(-1))
        exact[k] = lam
    add("lambda | exact float64 solve of the path's own (S, gamma)", lambda k: terr(exact[k], exact["o64"]))
    add("lambda | own PCG at its floor", lambda k: terr(st[k]["lambda"][:, 1:N + 1].reshape(B, -1), exact["o64"]))
    add("lambda | own PCG vs the exact solve of its OWN system", lambda k: terr(st[k]["lambda"][:, 1:N + 1].reshape(B, -1), exact[k]))
    add("dz", lambda k: terr(st[k]["dz"], st["o64"]["dz"]))
    add("XU after the step (max(1, .) scaling as in the parity tests)", lambda k: np.abs(full[k]["XU"].astype(np.float64) - full["o64"]["XU"]).max(axis=1) / np.maximum(1.0, np.abs(full["o64"]["XU"]).max(axis=1)))
    e_ho = np.abs(full["hip"]["XU"].astype(np.float64) - full["o32"]["XU"]).max(axis=1) / np.maximum(1.0, np.abs(full["o32"]["XU"]).max(axis=1))
    conds = [float(np.linalg.cond(dense_S(np.asarray(st["o64"]["S"][b], np.float64), N, nx))) for b in range(min(B, 2))]
    return {"plant": plant, "N": N, "B": B, "rows": rows, "hip_vs_o32_xu_max": float(e_ho.max()), "hip_vs_o32_xu_median": float(np.median(e_ho)),
            "pcg_iters": {k: [int(v) for v in st[k]["pcg_iters"]] for k in st}, "steps_equal_hip_o32": bool(np.array_equal(full["hip"]["ls_step_size"], full["o32"]["ls_step_size"])),
            "lambda_rel_of_exact_o64_pcg64": float(terr(st["o64"]["lambda"][:, 1:N + 1].reshape(B, -1), exact["o64"]).max()), "cond_S": conds}


def main():
    out = [run(*c) for c in [("indy7", 32, 16), ("iiwa14", 32, 8), ("iiwa14", 64, 8), ("iiwa14", 128, 8)]]
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "r04_stage_errors.json"), "w"), indent=1)
    with open(os.path.join(d, "r04_stage_errors.txt"), "w") as f:
        for r in out:
            f.write("%s N=%d B=%d   cond(S) ~ %s   HIP vs fp32 oracle after one iteration: max %.2e median %.2e, steps equal: %s\n" % (
                r["plant"], r["N"], r["B"], ", ".join("%.1e" % c for c in r["cond_S"]), r["hip_vs_o32_xu_max"], r["hip_vs_o32_xu_median"], r["steps_equal_hip_o32"]))
            f.write("  %-66s %10s %10s | %10s %10s\n" % ("error against the float64 oracle, per trajectory (median | max)", "HIP med", "HIP max", "o32 med", "o32 max"))
            for x in r["rows"]:
                f.write("  %-66s %10.2e %10.2e | %10.2e %10.2e\n" % (x["stage"], x["hip_median"], x["hip_max"], x["o32_median"], x["o32_max"]))
            f.write("\n")
    print(open(os.path.join(d, "r04_stage_errors.txt")).read())


if __name__ == "__main__":
    main()
