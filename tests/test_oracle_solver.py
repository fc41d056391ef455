"""The CPU oracle as a solver: regression against its committed golden outputs + the semantics of the reference's driver
(bsqp.cuh:103-197) that SURVEY.md Appendix A lists.  No GPU."""
import glob
import json
import os

import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from oracle.oracle import OracleSolver

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "oracle_*.npz")))


def _case(path):
    g = np.load(path)
    name = os.path.basename(path)[len("oracle_"):-4]
    plant, N, B = name.split("_")
    return g, plant, int(N[1:]), int(B[1:])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_oracle_reproduces_golden(path):
    g, plant, N, B = _case(path)
    p = json.loads(str(g["params"]))
    s = OracleSolver(plant, N, B, dt=float(g["dt"]), **p)
    s.set_f_ext_batch(g["in_f_ext"])
    s.setup_kkt(g["in_xu"], g["in_x_s"], g["in_ref"], 0.01)
    s.form_schur()
    for k in ("Q", "R", "q", "r", "A", "B", "c", "Qinv", "Rinv", "S", "Pinv", "gamma"):
        ref = g["st_" + k]
        np.testing.assert_allclose(s.buf(k), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
    out = s.solve(g["in_xu"], 0.01, g["in_x_s"], g["in_ref"])
    np.testing.assert_array_equal(out["ls_step_size"], g["out_ls_step_size"])
    np.testing.assert_array_equal(out["pcg_iters"], g["out_pcg_iters"])
    np.testing.assert_allclose(out["XU"], g["out_XU"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["final_merit"], g["out_final_merit"], rtol=1e-5)


def test_structure_of_blocks():
    """Facts the HIP path relies on: Q = blkdiag(dense qq, diagonal), R diagonal, S symmetric block-tridiagonal with zero padding blocks."""
    pr = fig8_problem("indy7", 8, 2, f_ext_std=3.0)
    p = dict(DEFAULT_SOLVER_PARAMS, vel_lim_cost=0.05, ctrl_lim_cost=0.02)
    s = OracleSolver("indy7", 8, 2, dt=0.01, **p)
    s.set_f_ext_batch(pr["f_ext"])
    s.setup_kkt(pr["xu"], pr["x_s"], pr["ref"], 0.01)
    s.form_schur()
    Q, R = s.buf("Q"), s.buf("R")
    assert np.all(Q[:, :, :6, 6:] == 0) and np.all(Q[:, :, 6:, :6] == 0)
    off = Q[:, :, 6:, 6:].copy()
    idx = np.arange(6)
    off[:, :, idx, idx] = 0
    assert np.all(off == 0)
    offr = R.copy()
    offr[:, :, idx, idx] = 0
    assert np.all(offr == 0)
    S = s.buf("S")  # [B, N, nx, 3nx] row-major block rows [left | main | right]
    assert np.all(S[:, 0, :, :12] == 0) and np.all(S[:, -1, :, 24:] == 0)
    for k in range(7):
        np.testing.assert_array_equal(S[:, k, :, 24:], np.swapaxes(S[:, k + 1, :, :12], 1, 2))  # right_k = left_{k+1}^T
    # terminal blocks: computed from knot N-2's state against knot N-1's reference with q_cost
    assert not np.allclose(Q[:, -1], Q[:, -2])
    q = s.buf("q")
    np.testing.assert_array_equal(q[:, -1, 6:], q[:, -2, 6:])  # velocity part identical: same state, no reference dependence


def test_solve_statistics_shapes_and_semantics():
    B, N = 3, 8
    pr = fig8_problem("indy7", N, B)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=4)
    s = OracleSolver("indy7", N, B, dt=0.01, **p)
    out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    assert out["XU"].shape == (B, 18 * N - 6)
    assert out["ls_num_iters"] == 4 and out["pcg_iters"].shape == (4, B) and out["ls_min_merit"].shape == (4, B)
    assert np.all(out["sqp_iters"] == 4)
    # ls_min_merit is the running best: non-increasing, starts below the initial merit where the first search succeeded
    mm = np.vstack([out["initial_merit"][None], out["ls_min_merit"]])
    assert np.all(np.diff(mm, axis=0) <= 0)
    st = out["ls_step_size"]
    assert np.all((st == -1) | ((st > 0) & (st <= 1)))
    assert np.all(np.diff(mm, axis=0)[st == -1] == 0)  # failed search leaves the merit unchanged
    # final merit is the merit of the returned trajectory recomputed with dz = 0
    fm = s.merit(out["XU"], pr["x_s"], pr["ref"], 0.01, num_alphas=1, zero_dz=True)[:, 0]
    np.testing.assert_allclose(out["final_merit"], fm, rtol=1e-6)
    # rho persists across solves, drho is reset (bsqp.cuh:189)
    assert np.all(s.buf("drho") == 1.0) and not np.all(s.buf("rho") == np.float32(0.01))
    s.reset_rho()
    assert np.all(s.buf("rho") == np.float32(0.01))
    # warm-started duals persist and change the next solve; reset_dual restores the first result
    out2 = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    assert not np.array_equal(out2["pcg_iters"], out["pcg_iters"])
    s.reset_dual(); s.reset_rho()
    out3 = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    np.testing.assert_array_equal(out3["XU"], out["XU"])


def test_converged_trajectories_trigger_early_exit():
    """PCG taking 0 iterations is the only convergence criterion and the loop breaks BEFORE the line search (bsqp.cuh:153,165)."""
    B, N = 2, 8
    pr = fig8_problem("indy7", N, B)
    # a huge pcg_tol makes |rho'| < 1e-6 + eps*|rho0| true after one iteration, never zero iterations -> no early exit
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3, pcg_tol=1e6)
    s = OracleSolver("indy7", N, B, dt=0.01, **p)
    out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    assert np.all(out["pcg_iters"] == 1) and out["iters_done"] == 3 and not out["kkt_converged"].any()
    # solve_ratio = 0 exits in the first iteration before any line search: xu unchanged
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3, solve_ratio=0.0)
    s = OracleSolver("indy7", N, B, dt=0.01, **p)
    out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    assert out["iters_done"] == 1 and out["ls_num_iters"] == 0 and np.all(out["sqp_iters"] == 1)
    np.testing.assert_array_equal(out["XU"], pr["xu"])
    np.testing.assert_allclose(out["final_merit"], out["initial_merit"], rtol=1e-6)


def test_batch_independence_and_per_trajectory_hyperparameters():
    N = 8
    pr = fig8_problem("iiwa14", N, 4, f_ext_std=2.0)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2)
    full = OracleSolver("iiwa14", N, 4, dt=0.01, **p)
    full.set_f_ext_batch(pr["f_ext"])
    rho = np.array([0.01, 0.1, 0.001, 0.05], np.float32)
    mu = np.array([10, 5, 20, 1], np.float32)
    full.set_rho_penalty_batch(rho); full.set_mu_batch(mu)
    o = full.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    for b in range(4):
        one = OracleSolver("iiwa14", N, 1, dt=0.01, **p)
        one.set_f_ext_batch(pr["f_ext"][b:b + 1])
        one.set_rho_penalty_batch(rho[b:b + 1]); one.set_mu_batch(mu[b:b + 1])
        ob = one.solve(pr["xu"][b:b + 1], 0.01, pr["x_s"][b:b + 1], pr["ref"][b:b + 1])
        np.testing.assert_array_equal(ob["XU"][0], o["XU"][b])
        np.testing.assert_array_equal(ob["final_merit"][0], o["final_merit"][b])


def test_sim_forward_uses_one_state_many_wrenches():
    s = OracleSolver("indy7", 8, 3, dt=0.01, **DEFAULT_SOLVER_PARAMS)
    f = np.zeros((3, 6), np.float32); f[1, 2] = 10.0; f[2, 4] = -5.0
    s.set_f_ext_batch(f)
    xk = np.concatenate([[-1.0, -0.1, 0.8, -0.1, 0.5, 0.0], np.zeros(6)]).astype(np.float32)
    out = s.sim_forward(xk, np.zeros(6, np.float32), 0.01)
    assert out.shape == (3, 12) and not np.allclose(out[0], out[1]) and not np.allclose(out[0], out[2])
    # type-2 rule: qd+ = qd + dt qdd, q+ = q + dt qd + dt^2/2 qdd  (integrator.cuh:34-37)
    qdd = (out[0, 6:] - xk[6:]) / 0.01
    np.testing.assert_allclose(out[0, :6], xk[:6] + 0.01 * xk[6:] + 0.5 * 0.01 ** 2 * qdd, atol=1e-6)


def test_per_trajectory_cost_weights_equal_one_solver_per_tuple():
    """Extension SURVEY 8(f)3 in the oracle: a batch whose rows carry their own cost weights reproduces, row by row and bit for bit,
    solvers constructed with those weights as scalars."""
    import numpy as np
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import HPARAM_COST_GRID, fig8_problem
    from oracle.oracle import OracleSolver
    N, B, dt = 8, 4, 0.01
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2)
    pr = fig8_problem("indy7", N, B)
    w = np.array([[g["q_cost"], g["qd_cost"], g["u_cost"], g["N_cost"], 0.01, 0.0, 1e-4 * (i % 2)] for i, g in
                  enumerate(HPARAM_COST_GRID[3:3 + B])], np.float32)
    one = OracleSolver("indy7", N, B, dt=dt, **p)
    one.set_cost_weights_batch(w)
    r1 = one.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    for i in range(B):
        pi = dict(p, q_cost=float(w[i, 0]), qd_cost=float(w[i, 1]), u_cost=float(w[i, 2]), N_cost=float(w[i, 3]), q_lim_cost=float(w[i, 4]),
                  vel_lim_cost=float(w[i, 5]), ctrl_lim_cost=float(w[i, 6]))
        s = OracleSolver("indy7", N, 1, dt=dt, **pi)
        ri = s.solve(pr["xu"][i:i + 1], dt, pr["x_s"][i:i + 1], pr["ref"][i:i + 1])
        np.testing.assert_array_equal(ri["XU"][0], r1["XU"][i])


@pytest.mark.parametrize("plant,N", [("iiwa14", 8), ("indy7", 32), ("iiwa14", 16)])
def test_mixed_batch_convergence_and_solve_ratio(plant, N):
    """bsqp.cuh:142-167 / pcg.cuh:29-32 where they act (tests/mixed_batch.py): a strict subset converged at entry (E, U rows), rows that converge
    later (P), rows that never do (F); solve_ratio values that exit in the first, in a later and in no iteration.  The same cases run on the
    device against this oracle in tests/test_convergence_gpu.py."""
    from mixed_batch import check_record_semantics, mixed_problem
    from oracle import oracle as O
    pr = mixed_problem(plant, N, ee=lambda p, q: O.ee(p, q)[0])
    kinds = pr["kinds"]
    B, iters = len(kinds), 6
    out = {}
    for ratio in (0.3, 0.5, 1.0):
        for f64 in (False, True):
            p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters, solve_ratio=ratio, pcg_tol=1e-8, max_pcg_iters=1000)
            s = OracleSolver(plant, N, B, dt=0.01, f64=f64, **p)
            s.set_f_ext_batch(pr["f_ext"]); s.set_cost_weights_batch(pr["w"])
            r = out[ratio, f64] = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
            check_record_semantics(r, B, ratio, iters)
            np.testing.assert_array_equal(r["pcg_iters_all"][0] == 0, np.array([k in "EU" for k in kinds]))
        a, b = out[ratio, False], out[ratio, True]     # fp32 and float64 agree on every counter and flag
        assert a["iters_done"] == b["iters_done"] and a["ls_num_iters"] == b["ls_num_iters"]
        np.testing.assert_array_equal(a["kkt_converged"], b["kkt_converged"])
        np.testing.assert_array_equal(a["pcg_iters_all"] == 0, b["pcg_iters_all"] == 0)
    r = out[0.3, False]
    assert r["iters_done"] == 1 and r["ls_num_iters"] == 0 and r["pcg_iters"].shape[0] == 0     # one PCG record more than line searches, truncated
    np.testing.assert_array_equal(r["XU"], pr["xu"])
    r = out[0.5, False]
    assert 2 <= r["iters_done"] < iters and r["ls_num_iters"] == r["iters_done"] - 1               # exit in a later iteration
    r = out[1.0, False]
    u = np.array([k == "U" for k in kinds])
    assert r["iters_done"] == iters and 0 < r["kkt_converged"].sum() < B
    assert np.all(r["ls_step_size"][0][u] == 1.0) and np.all(np.abs(r["XU"][u] - pr["xu"][u]).max(axis=1) > 0.1)   # converged at entry and still moved
    e = np.array([k == "E" for k in kinds])
    assert np.abs(r["XU"][e] - pr["xu"][e]).max() < 1e-5
