"""Pins the ORACLE's linear-algebra half (Schur complement, gamma signs, PCG, dz recovery) by mathematics instead of by reference
outputs (none exist: DESIGN.md section 3): in the float64 build of the same C source, with PCG run to its floor, the step dz must
satisfy the linearised equality constraints of the QP the blocks describe,

    dz_x0 + c_0 = 0,        dz_x,k+1 - A_k dz_x,k - B_k dz_u,k + c_k+1 = 0        (c_k+1 = x_k+1 - f(x_k,u_k), setup_kkt.cuh:75-98)

and lambda must solve S lambda = gamma.  A sign or index error anywhere between setup_kkt and compute_dz shows
up here as an O(1) residual; fp32-only effects do not (float64, residual ~1e-10).  No GPU."""
import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from oracle.oracle import OracleSolver


@pytest.mark.parametrize("plant,N,B,fstd", [("indy7", 8, 2, 0.0), ("iiwa14", 8, 2, 3.0), ("indy7", 32, 1, 2.0)])
def test_dz_satisfies_the_linearised_dynamics_f64(plant, N, B, fstd):
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1, pcg_tol=1e-16, max_pcg_iters=5000)
    pr = fig8_problem(plant, N, B, f_ext_std=fstd)
    xu = pr["xu"].astype(np.float64)
    # a non-trivial linearisation point: perturb the warm start so that defects c_k+1 are not zero
    rng = np.random.default_rng(5)
    xu += rng.normal(0, 0.05, xu.shape)
    o = OracleSolver(plant, N, B, dt=0.01, f64=True, **p)
    o.set_f_ext_batch(pr["f_ext"])
    o.setup_kkt(xu, pr["x_s"], pr["ref"], 0.01)
    A, Bm, c = o.buf("A"), o.buf("B"), o.buf("c")          # A[b,k,col,row], B[b,k,col,row] (col-major blocks)
    o.form_schur()
    S, gam = o.buf("S"), o.buf("gamma")
    Pinv = o.buf("Pinv")
    nx, nu = o.nx, o.nu
    ks = nx + nu

    def btd(M, v):   # block-tridiagonal product, block rows [left | main | right], vector padded by one block each side
        return np.stack([M[k] @ v[k:k + 3].reshape(-1) for k in range(N)])

    # the PCG of the oracle stops by pcg.cuh:96-141's rule |r^T P^-1 r| < 1e-6 + eps |rho_0|: with eps ~ 0 that is what its lambda satisfies
    o.pcg()
    lam = o.buf("lambda")
    for b in range(B):
        r = gam[b, 1:N + 1] - btd(S[b], lam[b])
        rp = np.zeros((N + 2, nx)); rp[1:N + 1] = r
        assert abs(float((r * btd(Pinv[b], rp)).sum())) < 1e-6 * 1.01
    # the EXACT solution of S lambda = gamma (dense float64 solve), handed to the oracle's dz recovery
    lam_x = np.zeros_like(lam)
    for b in range(B):
        Sd = np.zeros((N * nx, N * nx))
        for k in range(N):
            for j, kk in enumerate((k - 1, k, k + 1)):
                if 0 <= kk < N:
                    Sd[k * nx:(k + 1) * nx, kk * nx:(kk + 1) * nx] = S[b, k][:, j * nx:(j + 1) * nx]
        assert np.abs(Sd - Sd.T).max() < 1e-9 * np.abs(Sd).max()     # S is symmetric: right_k = left_{k+1}^T (schur_linsys.cuh:130-146)
        lam_x[b, 1:N + 1] = np.linalg.solve(Sd, gam[b, 1:N + 1].reshape(-1)).reshape(N, nx)
    o.set_lambda(lam_x)
    o.compute_dz()
    dz = o.buf("dz")
    # linearised dynamics
    scale = max(1.0, np.abs(dz).max())
    for b in range(B):
        d = dz[b]
        assert np.abs(d[:nx] + c[b, 0]).max() < 1e-8 * scale
        for k in range(N - 1):
            dxk, duk, dxn = d[k * ks:k * ks + nx], d[k * ks + nx:(k + 1) * ks], d[(k + 1) * ks:(k + 1) * ks + nx]
            Ak, Bk = A[b, k].T, Bm[b, k].T                 # -> [row, col]
            res = dxn - Ak @ dxk - Bk @ duk + c[b, k + 1]
            assert np.abs(res).max() < 1e-8 * scale, (b, k, np.abs(res).max())


def test_fp32_oracle_tracks_its_float64_build():
    """One SQP iteration, PCG at its floor: the fp32 oracle stays within a few 1e-4 of the float64 build of the same source and takes
    the same line-search steps (the intrinsic fp32 sensitivity of the problem: tools/sensitivity.py)."""
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1, pcg_tol=1e-9, max_pcg_iters=1000)
    pr = fig8_problem("indy7", 16, 4)
    out = []
    for f64 in (False, True):
        s = OracleSolver("indy7", 16, 4, dt=0.01, f64=f64, **p)
        out.append(s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"]))
    a, b = out
    np.testing.assert_array_equal(a["ls_step_size"], b["ls_step_size"])
    err = np.abs(a["XU"] - b["XU"]).max(axis=1) / np.maximum(1.0, np.abs(b["XU"]).max(axis=1))
    assert err.max() < 1e-3 and err.max() > 1e-7   # not bit-identical builds, and not far apart
