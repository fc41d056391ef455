"""The float64 build of the HIP path (libgato_hip_f64.so: the SAME kernel sources with double as the real type, the reference's
USE_DOUBLES of gato/settings.h:7-11) -- against the float64 oracle, as the arbiter of the fp32 path at the BASELINE sizes, and behind the
`BSQP_{B}_double` classes (python/bindings.cu:244-252)."""
import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem

pytestmark = pytest.mark.gpu
DT = 0.01


@pytest.mark.parametrize("plant,N,B,iters,tight", [("indy7", 8, 2, 1, True), ("indy7", 32, 5, 3, True), ("iiwa14", 16, 3, 3, True), ("indy7", 64, 2, 2, True),
                                                   ("iiwa14", 128, 2, 2, True), ("indy7", 32, 6, 1, False), ("iiwa14", 64, 3, 1, False), ("indy7", 128, 2, 3, True), ("indy7", 32, 4, 6, True)])
def test_float64_build_equals_the_float64_oracle(plant, N, B, iters, tight):
    """Two independent implementations of the path (kernels.hpp: fused Schur + PCG, pair form, symmetric-storage PCG, fused step ... and
    oracle/gato_oracle.c) in double: every decision identical, iterates to 1e-9 -- what separates the fp32 builds of the two is rounding.
    (Several iterations at the DEFAULT PCG tolerance are not comparable even in double: the exit test is a discontinuity, a count that differs
    by one moves lambda within the tolerance and the trajectories part; those cases run one iteration.)"""
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters)
    if tight:
        p.update(pcg_tol=1e-9, max_pcg_iters=1000)
    pr = fig8_problem(plant, N, B, f_ext_std=2.0)
    nat = NativeSolver(plant, N, B, f64=True, dt=DT, **p)
    o64 = OracleSolver(plant, N, B, dt=DT, f64=True, **p)
    for s in (nat, o64):
        s.set_f_ext_batch(pr["f_ext"])
    rg = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    ro = o64.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    assert rg["XU"].dtype == np.float64
    np.testing.assert_array_equal(rg["ls_step_size"], ro["ls_step_size"])
    np.testing.assert_array_equal(rg["sqp_iters"], ro["sqp_iters"])
    assert np.abs(rg["pcg_iters"] - ro["pcg_iters"]).max() <= (0 if tight else 1)
    assert np.abs(rg["XU"] - ro["XU"]).max() <= 1e-9 * np.abs(ro["XU"]).max()
    for k in ("final_merit", "initial_merit", "ls_min_merit"):
        assert np.abs(rg[k] - ro[k]).max() <= 1e-9 * np.abs(ro[k]).max(), k
    assert np.abs(nat.read("rho") - o64.buf("rho")).max() <= 1e-12
    nat.stage("kkt", pr["xu"], DT, pr["x_s"], pr["ref"])
    nat.stage("schur", pr["xu"], DT, pr["x_s"], pr["ref"])
    o64.setup_kkt(pr["xu"], pr["x_s"], pr["ref"], DT)
    o64.form_schur()
    for name in ("S", "Pinv", "gamma"):
        a, b = nat.read(name).reshape(o64.buf(name).shape), o64.buf(name)
        assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), name


@pytest.mark.parametrize("plant,N,B", [("indy7", 32, 1024), ("iiwa14", 128, 256), ("iiwa14", 64, 512)])
def test_fp32_path_against_its_float64_build_at_full_size(plant, N, B):
    """BASELINE configurations C2, C3 and C5's per-GPU shard, EVERY trajectory: one SQP iteration with PCG at its floor on the fp32
    library and on its float64 build (the arbiter: equal to the float64 oracle, test above).  fp32 cannot do better than this on these
    systems (the fp32 oracle is the same distance away): steps equal on >= 99 %, iterates of those within 2e-3 at
    worst, 6e-4 at the 99th percentile, 2e-4 in the median (relative to the trajectory's largest entry)."""
    from gato_amd._lib import NativeSolver
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1, pcg_tol=1e-9, max_pcg_iters=1000)
    pr = fig8_problem(plant, N, B)
    out = {}
    for f64 in (False, True):
        out[f64] = NativeSolver(plant, N, B, f64=f64, dt=DT, **p).solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    a, b = out[False], out[True]
    same = np.all(a["ls_step_size"] == b["ls_step_size"].astype(np.float32), axis=0)
    e = np.abs(a["XU"].astype(np.float64) - b["XU"]).max(axis=1) / np.maximum(1.0, np.abs(b["XU"]).max(axis=1))
    assert same.mean() >= 0.99, same.sum()
    assert e[same].max() < 2e-3 and np.quantile(e[same], 0.99) < 6e-4 and np.median(e[same]) < 2e-4, (e[same].max(), np.quantile(e[same], 0.99), np.median(e))
    assert np.abs(a["initial_merit"] - b["initial_merit"]).max() <= 1e-5 * np.abs(b["initial_merit"]).max()
    assert np.all(np.isfinite(a["XU"]))


def test_double_classes_of_the_modules():
    """`BSQP_{B}_double` (what a USE_DOUBLES build of the reference registers, bindings.cu:244-252): the compiled float64 binding gives the
    bits of the ctypes path on libgato_hip_f64.so, float64 arrays in and out."""
    import gato_amd.bsqp.bsqpN16_indy7 as m
    from gato_amd._lib import NativeSolver
    N, B = 16, 4
    pr = fig8_problem("indy7", N, B, f_ext_std=1.0)
    p = DEFAULT_SOLVER_PARAMS
    args = [DT, 3, p["kkt_tol"], p["max_pcg_iters"], p["pcg_tol"], p["solve_ratio"], p["mu"], p["q_cost"], p["qd_cost"], p["u_cost"], p["N_cost"],
            p["q_lim_cost"], p["vel_lim_cost"], p["ctrl_lim_cost"], p["rho"]]
    s = m.BSQP_4_double(*args)
    s.set_f_ext_batch(pr["f_ext"].astype(np.float64))
    r = s.solve(pr["xu"].astype(np.float64), DT, pr["x_s"].astype(np.float64), pr["ref"].astype(np.float64))
    nat = NativeSolver("indy7", N, B, f64=True, dt=DT, **dict(p, max_sqp_iters=3))
    nat.set_f_ext_batch(pr["f_ext"])
    q = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    assert r["XU"].dtype == np.float64
    np.testing.assert_array_equal(r["XU"], q["XU"])
    np.testing.assert_array_equal(r["pcg_iters"], q["pcg_iters"])
    np.testing.assert_array_equal(r["final_merit"], q["final_merit"])
    f = m.BSQP_4_float(*args)
    f.set_f_ext_batch(pr["f_ext"])
    rf = f.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    assert rf["XU"].dtype == np.float32 and np.median(np.abs(rf["XU"] - r["XU"])) < 1e-3


def test_random_configurations_in_double():
    """The randomised sweep of tools/fuzz_parity.py where it is crisp: both implementations in double.  Plant, horizon, batch, time step,
    wrench, all seven cost weights (barrier terms included), rho and mu at random; two free iterations with PCG at its floor: same steps,
    iteration counts within one, iterates to 1e-8 (1e-4 where the PCG needs more than 200 iterations even in double) -- no tolerance to hide an algebra error in a rarely taken branch."""
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    rng = np.random.default_rng(11)
    worst = 0.0
    for case in range(24):
        plant = str(rng.choice(["indy7", "iiwa14"]))
        N = int(rng.choice([4, 8, 16, 32, 64]))
        B = int(rng.integers(1, 6))
        dt = float(rng.choice([0.005, 0.01, 0.02]))
        p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2, pcg_tol=1e-10, max_pcg_iters=2000, rho=float(10 ** rng.uniform(-3, -1)),
                 mu=float(rng.choice([1.0, 10.0, 50.0])), q_cost=float(rng.choice([0.5, 2.0, 10.0])), qd_cost=float(10 ** rng.uniform(-4, -1)),
                 u_cost=float(10 ** rng.uniform(-7, -5)), N_cost=float(rng.choice([10.0, 50.0, 100.0])), q_lim_cost=float(rng.choice([0.0, 0.01])),
                 vel_lim_cost=float(rng.choice([0.0, 1e-3])), ctrl_lim_cost=float(rng.choice([0.0, 1e-4])))
        pr = fig8_problem(plant, N, B, seed=int(rng.integers(0, 1000)), dt=0.01, f_ext_std=float(rng.choice([0.0, 3.0])))
        nat = NativeSolver(plant, N, B, f64=True, dt=dt, **p)
        o64 = OracleSolver(plant, N, B, dt=dt, f64=True, **p)
        for s in (nat, o64):
            s.set_f_ext_batch(pr["f_ext"])
        rg = nat.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
        ro = o64.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
        tag = (case, plant, N, B, dt, p)
        np.testing.assert_array_equal(rg["ls_step_size"], ro["ls_step_size"], err_msg=str(tag))
        assert np.abs(rg["pcg_iters"] - ro["pcg_iters"]).max() <= 1, tag   # an exit test crossed one iteration apart, at several hundred
        e = float(np.abs(rg["XU"] - ro["XU"]).max() / np.abs(ro["XU"]).max())
        worst = max(worst, e)
        # a PCG that needs several hundred iterations in double sits on cond(S) ~ 1e10: its solution carries 1e-10 x cond of rounding
        hard = int(ro["pcg_iters"].max()) > 200
        assert e <= (1e-4 if hard else 1e-8), (tag, e)
        if not hard:   # (the merit multiplies a 1e-5 iterate difference by mu |d defect / d x|: not a measure there)
            assert np.abs(rg["final_merit"] - ro["final_merit"]).max() <= 1e-8 * np.abs(ro["final_merit"]).max(), tag
    assert worst > 0.0   # two implementations, not one compared with itself
