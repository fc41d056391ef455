"""The oracle's rigid-body dynamics against the identities the reference's algorithms imply (no GPU).

The reference (CUDA) cannot run here, so the dynamics of the oracle are pinned by mathematics instead: with the tables proven equal
to the reference's literals (test_robot_tables.py), RNEA / direct M^-1 / analytic gradients / FK are unique functions of (q,qd,u,f_ext).
The float64 build of the same C source (oracle/Makefile target f64) makes the identities hold to ~1e-9.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(O.__file__))
NQ = {0: 6, 1: 7}


@pytest.fixture(scope="module")
def L64():
    path = os.path.join(HERE, "libgato_oracle_f64.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", HERE, "-s", "libgato_oracle_f64.so"])
    return C.CDLL(path)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fd(L, pl, q, qd, u, fe):
    out = np.zeros(NQ[pl])
    L.orc_fd(pl, _p(q), _p(qd), _p(u), _p(fe), _p(out))
    return out


@pytest.mark.parametrize("pl", [0, 1])
def test_forward_inverse_dynamics_consistency_f64(L64, pl):
    rng = np.random.default_rng(3)
    nq = NQ[pl]
    for _ in range(5):
        q, qd, u, fe = rng.uniform(-2, 2, nq), rng.uniform(-2, 2, nq), rng.uniform(-20, 20, nq), rng.normal(0, 5, 6)
        qdd = _fd(L64, pl, q, qd, u, fe)
        c = np.zeros(nq)
        L64.orc_rnea(pl, _p(q), _p(qd), _p(qdd), _p(fe), _p(c))
        np.testing.assert_allclose(c, u, atol=1e-10)  # RNEA(FD(u)) == u
        # mass matrix from RNEA columns (gravity and wrench cancel in the difference) is symmetric and inverted by direct_minv
        z = np.zeros(nq)
        g = np.zeros(nq)
        L64.orc_rnea(pl, _p(q), _p(z), _p(z), _p(fe), _p(g))
        M = np.zeros((nq, nq))
        for i in range(nq):
            e = np.zeros(nq); e[i] = 1
            col = np.zeros(nq)
            L64.orc_rnea(pl, _p(q), _p(z), _p(e), _p(fe), _p(col))
            M[:, i] = col - g
        Mi = np.zeros(nq * nq)
        L64.orc_minv(pl, _p(q), _p(Mi))
        Mi = Mi.reshape(nq, nq).T
        np.testing.assert_allclose(M, M.T, atol=1e-12)
        np.testing.assert_allclose(M @ Mi, np.eye(nq), atol=1e-9)
        assert np.all(np.linalg.eigvalsh(M) > 0)


@pytest.mark.parametrize("pl", [0, 1])
def test_analytic_gradients_equal_central_differences_f64(L64, pl):
    rng = np.random.default_rng(4)
    nq = NQ[pl]
    h = 1e-6
    for _ in range(3):
        q, qd, u, fe = rng.uniform(-2, 2, nq), rng.uniform(-2, 2, nq), rng.uniform(-20, 20, nq), rng.normal(0, 5, 6)
        qdd = np.zeros(nq)
        D = np.zeros(3 * nq * nq)
        L64.orc_fd_grad(pl, _p(q), _p(qd), _p(u), _p(fe), _p(qdd), _p(D))
        D = D.reshape(3 * nq, nq).T
        Dn = np.zeros((nq, 3 * nq))
        for i in range(nq):
            e = np.zeros(nq); e[i] = h
            Dn[:, i] = (_fd(L64, pl, q + e, qd, u, fe) - _fd(L64, pl, q - e, qd, u, fe)) / (2 * h)
            Dn[:, nq + i] = (_fd(L64, pl, q, qd + e, u, fe) - _fd(L64, pl, q, qd - e, u, fe)) / (2 * h)
            Dn[:, 2 * nq + i] = (_fd(L64, pl, q, qd, u + e, fe) - _fd(L64, pl, q, qd, u - e, fe)) / (2 * h)
        assert np.abs(D - Dn).max() / np.abs(Dn).max() < 1e-7
        # FK Jacobian
        e0 = np.zeros(3); J = np.zeros(3 * nq)
        L64.orc_ee(pl, _p(q), _p(e0), _p(J))
        J = J.reshape(nq, 3).T
        Jn = np.zeros((3, nq))
        for i in range(nq):
            e = np.zeros(nq); e[i] = h
            ep, em, dummy = np.zeros(3), np.zeros(3), np.zeros(3 * nq)
            qp, qm = q + e, q - e
            L64.orc_ee(pl, _p(qp), _p(ep), _p(dummy))
            L64.orc_ee(pl, _p(qm), _p(em), _p(dummy))
            Jn[:, i] = (ep - em) / (2 * h)
        np.testing.assert_allclose(J, Jn, atol=1e-8)
        assert np.allclose(J[:, -1], 0)  # the EE is the origin of the last joint frame (indy7_grid.cuh:1886 "TODO: ADD OFFSETS")


@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_f32_matches_f64(L64, plant):
    pl = O.PLANTS[plant]
    nq = NQ[pl]
    rng = np.random.default_rng(5)
    q, qd, u, fe = rng.uniform(-1.5, 1.5, nq), rng.uniform(-1, 1, nq), rng.uniform(-10, 10, nq), rng.normal(0, 5, 6)
    qdd32, D32 = O.fd_grad(plant, q, qd, u, fe)
    q_, qd_, u_, fe_ = [np.float32(x).astype(np.float64) for x in (q, qd, u, fe)]
    qdd = np.zeros(nq); D = np.zeros(3 * nq * nq)
    L64.orc_fd_grad(pl, _p(q_), _p(qd_), _p(u_), _p(fe_), _p(qdd), _p(D))
    D = D.reshape(3 * nq, nq).T
    assert np.abs(qdd32 - qdd).max() / max(1, np.abs(qdd).max()) < 5e-4
    assert np.abs(D32 - D).max() / np.abs(D).max() < 5e-4


def test_ee_known_configuration():
    # indy7 at q = 0: z = 0.0775+0.222+0.45+0.267+0.083+0.168 (joint origins, indy7.urdf:208-248), no tool offset
    e, J = O.ee("indy7", np.zeros(6))
    np.testing.assert_allclose(e, [0.0, -0.1865, 1.2675], atol=1e-6)
    # the first point of the fig-8 reference is reachable workspace-wise (|e| sanity)
    assert 0.5 < np.linalg.norm(e) < 1.6


def test_gauss_jordan_matches_numpy_inverse():
    rng = np.random.default_rng(0)
    for n in (6, 7, 12, 14):
        A = rng.normal(size=(n, n))
        M = A @ A.T + n * np.eye(n)
        for form in (False, True):
            inv = O.gj_inverse(M, one_matrix_form=form)
            np.testing.assert_allclose(inv @ M, np.eye(n), atol=2e-5)
