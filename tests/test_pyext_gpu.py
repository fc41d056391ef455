"""The compiled Python binding (gato_amd._gato_ext, pybind11 over the C ABI) on the device: same bits as the ctypes back door the
parity tests use, the reference's result-dict surface (python/bindings.cu:96-145), and the device-side hypothesis selection."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402

ARGS = ["dt", "max_sqp_iters", "kkt_tol", "max_pcg_iters", "pcg_tol", "solve_ratio", "mu", "q_cost", "qd_cost", "u_cost", "N_cost", "q_lim_cost",
        "vel_lim_cost", "ctrl_lim_cost", "rho"]


@pytest.mark.parametrize("plant,N,B", [("indy7", 32, 8), ("iiwa14", 16, 3)])
def test_compiled_class_equals_ctypes_path(plant, N, B):
    import importlib
    from gato_amd._lib import NativeSolver
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3, dt=0.01)
    pr = fig8_problem(plant, N, B, f_ext_std=2.0)
    mod = importlib.import_module("gato_amd.bsqp.bsqpN%d_%s" % (N, plant))
    s = getattr(mod, "BSQP_%d_float" % B)(*[p[k] for k in ARGS])
    s.set_f_ext_batch(pr["f_ext"])
    r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    c = NativeSolver(plant, N, B, **p)
    c.set_f_ext_batch(pr["f_ext"])
    rc = c.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    for k in ("XU", "sqp_iters", "kkt_converged", "final_merit", "initial_merit", "pcg_iters", "ls_min_merit", "ls_step_size", "pcg_times_us"):
        np.testing.assert_array_equal(r[k], rc[k], err_msg=k)
    # dtypes and shapes of PyBSQP::solve's dict (bindings.cu:96-145)
    assert r["XU"].dtype == np.float32 and r["XU"].shape == (B, c.traj) and r["sqp_iters"].dtype == np.int32 and r["kkt_converged"].dtype == np.int32
    assert r["pcg_iters"].shape == (3, B) and r["pcg_iters"].dtype == np.int32 and r["ls_min_merit"].shape == (3, B) and r["ls_num_iters"] == 3
    assert isinstance(r["sqp_time_us"], float) and r["sqp_time_us"] > 0 and r["pcg_times_us"].shape == (3,)
    # state semantics through the compiled class: lambda persists, reset restores
    r2 = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    assert not np.array_equal(r2["pcg_iters"], r["pcg_iters"])
    s.reset_dual(); s.reset_rho()
    r3 = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    np.testing.assert_array_equal(r3["XU"], r["XU"])
    with pytest.raises(ValueError):
        s.set_f_ext_batch(np.zeros((B, 5), np.float32))      # wrong size: rejected before it reaches the C ABI
    np.testing.assert_array_equal(s.sim_forward(pr["x_s"][0], np.zeros(c.nu, np.float32), 0.01), c.sim_forward(pr["x_s"][0], np.zeros(c.nu), 0.01))


@pytest.mark.parametrize("plant,B", [("indy7", 37), ("iiwa14", 1024), ("indy7", 4)])
def test_select_best_against_oracle(plant, B):
    """gato_select_best = sim_forward + per-hypothesis error + arg-min in one launch (mpc_controller.py:294-309) against the
    oracle's sim_forward followed by numpy."""
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    nat = NativeSolver(plant, 8, B)
    orc = OracleSolver(plant, 8, B)
    rng = np.random.default_rng(B)
    f = rng.normal(0, 6.0, (B, 6)).astype(np.float32)
    nat.set_f_ext_batch(f); orc.set_f_ext_batch(f)
    nx, nu = nat.nx, nat.nu
    xk = np.concatenate([rng.uniform(-0.6, 0.6, nu), rng.uniform(-0.3, 0.3, nu)]).astype(np.float32)
    uk = rng.uniform(-4, 4, nu).astype(np.float32)
    xo = orc.sim_forward(xk, uk, 0.008)
    truth = B // 3
    x_meas = (xo[truth].astype(np.float64) + rng.normal(0, 1e-7, nx)).astype(np.float32)
    best, err = nat.select_best(xk, uk, x_meas, 0.008)
    eo = np.linalg.norm(xo.astype(np.float32) - x_meas[None, :], axis=1)
    assert best == int(np.argmin(eo)) == truth
    np.testing.assert_allclose(err, eo, rtol=2e-4, atol=2e-6)   # the true hypothesis' error is rounding noise of ~1e-7
    # twice in a row: the completion counter of the selection kernel resets itself
    best2, err2 = nat.select_best(xk, uk, x_meas, 0.008)
    assert best2 == best and np.array_equal(err2, err)
    # ties resolve to the first index like np.argmin
    nat.set_f_ext_batch(np.zeros((B, 6), np.float32))
    b0, e0 = nat.select_best(xk, uk, x_meas, 0.008)
    assert b0 == 0 and np.all(e0 == e0[0])
