"""gato_amd.bsqp.{common,config} against outputs of the reference's own Python (golden fixture made by tools/make_golden.py)."""
import json
import os

import numpy as np

from gato_amd.bsqp import common, config
from gato_amd.bsqp.workloads import fig8_problem

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_python.npz")


def test_figure8_matches_reference_python():
    g = np.load(GOLD)
    mine = common.figure8(0.01, **config.FIG8_DEFAULT_PARAMS)
    assert mine.shape == g["fig8"].shape == (18000,)
    np.testing.assert_allclose(mine, g["fig8"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(mine.astype(np.float32), g["fig8"].astype(np.float32))  # bit-for-bit after the float32 cast
    np.testing.assert_allclose(mine[:6], [-0.35355339, 0.35355339, 0.8, 0, 0, 0], atol=1e-8)
    mine_b = common.figure8(0.02, A_x=0.3, A_z=0.2, offset=[0.1, 0.4, 0.5], period=4, cycles=2, theta=0.3)
    np.testing.assert_allclose(mine_b, g["fig8_b"], rtol=0, atol=1e-15)


def test_warm_start_and_configs_match_reference_python():
    g = np.load(GOLD)
    np.testing.assert_array_equal(common.initialize_warm_start(np.arange(12, dtype=float), 5, 12, 6), g["warm_start"])
    np.testing.assert_array_equal(config.INDY7_START_CONFIGS["ready"], g["indy7_ready"])
    assert config.DEFAULT_SOLVER_PARAMS == json.loads(str(g["default_solver_params"]))
    assert config.PICKPLACE_SOLVER_PARAMS == json.loads(str(g["pickplace_solver_params"]))
    assert list(g["standard_batch_sizes"]) == config.STANDARD_BATCH_SIZES
    ref_f8 = json.loads(str(g["fig8_default_params"]))
    for k, v in config.FIG8_DEFAULT_PARAMS.items():
        assert np.allclose(v, ref_f8[k])


def test_fig8_problem_shapes_and_sharding():
    pr = fig8_problem("indy7", 32, 6, seed=0)
    assert pr["xu"].shape == (6, 570) and pr["x_s"].shape == (6, 12) and pr["ref"].shape == (6, 192) and pr["f_ext"].shape == (6, 6)
    assert pr["xu"].dtype == np.float32
    np.testing.assert_array_equal(pr["xu"][:, :12], pr["x_s"])
    np.testing.assert_array_equal(pr["xu"][:, 18:30], pr["x_s"])       # warm start repeats x_s
    assert np.all(pr["xu"][:, 12:18] == 0)
    # rank-sharded generation reproduces the rows of the global problem
    a = fig8_problem("indy7", 32, 3, seed=0, batch_offset=3)
    for k in pr:
        np.testing.assert_array_equal(a[k], pr[k][3:])
    # distinct phases
    assert not np.array_equal(pr["ref"][0], pr["ref"][1])
    pi = fig8_problem("iiwa14", 16, 2, f_ext_std=5.0)
    assert pi["xu"].shape == (2, 21 * 16 - 7) and np.abs(pi["f_ext"]).max() > 0


def test_hparam_problem_c5():
    """C5: per-shard cost tuple, per-trajectory rho on a log grid, constant goal, zero start."""
    from gato_amd.bsqp.workloads import HPARAM_COST_GRID, hparam_problem
    assert len(HPARAM_COST_GRID) == 24 and HPARAM_COST_GRID[0] == dict(q_cost=10.0, qd_cost=1e-1, u_cost=1e-6, N_cost=100.0)
    a = hparam_problem("iiwa14", 64, 6, shard=3)
    b = hparam_problem("iiwa14", 64, 6, shard=3)
    c = hparam_problem("iiwa14", 64, 6, shard=4)
    assert a["xu"].shape == (6, 21 * 64 - 7) and a["ref"].shape == (6, 6 * 64) and a["dt"] == 0.05
    np.testing.assert_array_equal(a["ref"], b["ref"])
    assert not np.array_equal(a["ref"], c["ref"]) and a["params"]["qd_cost"] != c["params"]["qd_cost"] or a["params"] != c["params"]
    r = a["ref"].reshape(6, 64, 6)
    assert np.all(r[:, :, :3] == r[:, :1, :3]) and np.all(r[:, :, 3:] == 0) and np.all(a["x_s"] == 0) and np.all(a["xu"] == 0)
    assert np.all(np.abs(r[:, 0, :2]) <= 0.8) and np.all((r[:, 0, 2] >= 0.2) & (r[:, 0, 2] <= 0.8))
    np.testing.assert_allclose(a["rho"], 10.0 ** (-8 + 9 * (np.arange(6) + 1) / 513.0), rtol=1e-6)
    assert a["params"]["mu"] == 1.0 and a["params"]["pcg_tol"] == 1e-3 and a["params"]["max_sqp_iters"] == 10


def test_oracle_solves_hparam_problem():
    """The C5 settings drive the oracle end to end (dt = 0.05, mu = 1, per-trajectory rho, other cost weights)."""
    from gato_amd.bsqp.workloads import hparam_problem
    from oracle.oracle import OracleSolver
    pr = hparam_problem("iiwa14", 16, 3, shard=1)
    p = dict(pr["params"], max_sqp_iters=3)
    o = OracleSolver("iiwa14", 16, 3, dt=pr["dt"], **p)
    o.set_rho_penalty_batch(pr["rho"])
    r = o.solve(pr["xu"], pr["dt"], pr["x_s"], pr["ref"])
    assert np.all(np.isfinite(r["XU"])) and np.all(r["final_merit"] <= r["initial_merit"])
