"""A-priori rounding-error bounds for the two ill-conditioned stage maps of an SQP iteration (test infrastructure, numpy float64).

On the hyper-parameter sweep (BASELINE config C5: R^-1 = 1 / u_cost = 1e6 .. 1e7, cond(theta + rho I) ~ 4e9) the right-hand side gamma
(schur_linsys.cuh:81,121-128) and the step dz (schur_linsys.cuh:316-431) are sums with cancellation: dz_u = -R^-1 (r + B^T lambda) is 1e7
times a 14-term dot product that cancels to 1e-4 of its terms.  Two correct fp32 evaluations of the SAME formula on the SAME inputs (another
summation order, an fma where the other rounds twice) then differ by percents of max|dz| -- the fp32 oracle is up to 3.7e-2 from its own float64
build there -- so a fixed 1e-5 cannot be the parity statement.  The statement that CAN be made, deterministically and per component, is the
textbook forward-error bound of the formula (Higham, Accuracy and Stability of Numerical Algorithms, section 3.1: an n-term dot product
evaluated in any order, with or without fma, satisfies |fl(x.y) - x.y| <= gamma_n |x|.|y|, gamma_n ~ n u, u = 2^-24):

    |dz_computed - dz_exact(inputs)|       <=  K u  |Qinv| (|q| + |lambda_k| + |A^T| |lambda_k+1|)          (state rows;  K = 2 nx + 4)
                                           <=  K u  |Rinv| (|r| + |B^T| |lambda_k+1|)                       (control rows)
    |gamma_computed - gamma_exact(inputs)| <=  K u  (|c| + |Qinv_k+1||q_k+1| + (|A||Qinv_k|)|q_k| + (|B||Rinv_k|)|r_k|)     (K = 2 nx + 6)

`exact` = the same formula in float64 on the path's OWN fp32 inputs (the HIP path's device buffers, or the oracle's).  Every function takes the
reference's memory layout (column-major blocks: X[b, k, c, r] = X_k(r, c), linalg.cuh:545-672) and returns (exact, bound) with `bound` WITHOUT
the factor K u."""
import numpy as np

U = 2.0 ** -24


def _m(x):
    """stored column-major blocks -> mathematical matrices, float64"""
    return np.swapaxes(np.asarray(x, np.float64), -1, -2)


def dz_exact_and_bound(A, Bm, Qinv, Rinv, q, r, lam):
    """A, Qinv [B,N,nx,nx]; Bm [B,N,nu,nx]; Rinv [B,N,nu,nu]; q [B,N,nx]; r [B,N,nu] (the LINEARISATION's q, r -- before computeDz overwrites them
    with the residuals); lam [B,N+2,nx].  Returns dz, bound as [B, (nx+nu) N - nu]."""
    A, Bm, Qi, Ri = _m(A), _m(Bm), _m(Qinv), _m(Rinv)     # A_k (nx x nx), B_k (nx x nu)
    q, r, lam = (np.asarray(a, np.float64) for a in (q, r, lam))
    Bsz, N, nx = q.shape
    nu = r.shape[2]
    l1, l2 = lam[:, 1:N + 1], lam[:, 2:N + 2]
    live = (np.arange(N) < N - 1)[None, :, None]                              # knot N-1 has no successor and no control
    At_l = np.einsum("bkji,bkj->bki", A, l2) * live                           # A^T lambda_{k+2}
    aAt_l = np.einsum("bkji,bkj->bki", np.abs(A), np.abs(l2)) * live
    res = q - (l1 - At_l)
    bres = np.abs(q) + np.abs(l1) + aAt_l
    dzx = -np.einsum("bkij,bkj->bki", Qi, res)
    bx = np.einsum("bkij,bkj->bki", np.abs(Qi), bres)
    su = r + np.einsum("bkji,bkj->bki", Bm, l2)                               # r + B^T lambda_{k+2}
    bsu = np.abs(r) + np.einsum("bkji,bkj->bki", np.abs(Bm), np.abs(l2))
    dzu = -np.einsum("bkij,bkj->bki", Ri, su)
    bu = np.einsum("bkij,bkj->bki", np.abs(Ri), bsu)
    out, bnd = (np.concatenate([x, u_], axis=2).reshape(Bsz, -1)[:, : (nx + nu) * N - nu] for x, u_ in ((dzx, dzu), (bx, bu)))
    return out, bnd


def gamma_exact_and_bound(A, Bm, Qinv, Rinv, q, r, c):
    """gamma [B,N+2,nx] (padded like the reference's, rows 0 and N+1 zero) and its bound; c [B,N,nx]"""
    A, Bm, Qi, Ri = _m(A), _m(Bm), _m(Qinv), _m(Rinv)
    q, r, c = (np.asarray(a, np.float64) for a in (q, r, c))
    Bsz, N, nx = q.shape
    g = np.zeros((Bsz, N + 2, nx))
    b = np.zeros((Bsz, N + 2, nx))
    g[:, 1] = c[:, 0] - np.einsum("bij,bj->bi", Qi[:, 0], q[:, 0])
    b[:, 1] = np.abs(c[:, 0]) + np.einsum("bij,bj->bi", np.abs(Qi[:, 0]), np.abs(q[:, 0]))
    k = slice(0, N - 1)
    k1 = slice(1, N)
    phi = np.einsum("bkij,bkjl->bkil", A[:, k], Qi[:, k])
    aphi = np.einsum("bkij,bkjl->bkil", np.abs(A[:, k]), np.abs(Qi[:, k]))
    BR = np.einsum("bkij,bkjl->bkil", Bm[:, k], Ri[:, k])
    aBR = np.einsum("bkij,bkjl->bkil", np.abs(Bm[:, k]), np.abs(Ri[:, k]))
    gg = -c[:, k1] + np.einsum("bkij,bkj->bki", Qi[:, k1], q[:, k1]) - np.einsum("bkij,bkj->bki", phi, q[:, k]) - np.einsum("bkij,bkj->bki", BR, r[:, k])
    g[:, 2:N + 1] = -gg
    b[:, 2:N + 1] = (np.abs(c[:, k1]) + np.einsum("bkij,bkj->bki", np.abs(Qi[:, k1]), np.abs(q[:, k1])) + np.einsum("bkij,bkj->bki", aphi, np.abs(q[:, k]))
                     + np.einsum("bkij,bkj->bki", aBR, np.abs(r[:, k])))
    return g, b


def ratio(computed, exact, bound, K):
    """max over the components of |computed - exact| / (K u bound): <= 1 is the theorem (components whose bound is 0 must be exact)"""
    d = np.abs(np.asarray(computed, np.float64).reshape(exact.shape) - exact)
    tiny = K * U * bound
    assert np.all(d[tiny == 0] == 0)
    return float((d[tiny > 0] / tiny[tiny > 0]).max())
