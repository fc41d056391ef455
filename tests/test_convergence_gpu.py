"""The reference's ONLY stopping rule, on the device, against the oracle where it acts (bsqp.cuh:142-167, pcg.cuh:29-32).

"Converged" = PCG took 0 iterations (bsqp.cuh:153; kkt_tol is unused).  A converged trajectory skips PCG in later iterations (pcg.cuh:29-32) but
still gets dz / merit / line search and keeps moving; every trajectory's sqp_iters counts every executed iteration; the loop breaks BEFORE the
line search once num_solved >= B * solve_ratio (bsqp.cuh:165), which leaves one more PCG record than line searches -- the record the
binding truncates (bindings.cu:111-128).  Here that host loop is device logic (Ctrl / num_solved in kernels.hpp), so it is pinned on a batch
in which a STRICT SUBSET is converged at entry (tests/mixed_batch.py: wrench-held equilibria E, converged-but-moving rows U, rows P that
converge in later iterations, fig-8 rows F that never do), for solve_ratio values that exit in the first, in a later and in no iteration.

fp32 HIP path vs fp32 oracle: all counters, the flags and the converged-in-iteration record of the E / U / P rows (and of the fig-8 rows while
fp32 keeps two implementations on one branch), the first line searches of the rows that have a decision to make; rows resting at their
optimum compare merits that differ by rounding only: there the ITERATES are compared, to 5e-5.
float64 HIP build vs float64 oracle: EVERYTHING, every row, every iteration -- exactly / to 1e-9."""
import os

import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from mixed_batch import check_record_semantics as _check_record_semantics, mixed_problem

pytestmark = pytest.mark.gpu
DT = 0.01
PCG = dict(pcg_tol=1e-8, max_pcg_iters=1000)
ITERS = 6
# (plant, N): indy7 N=32 = fused Schur + PCG (pcgc FUSE, pair form at this batch size), iiwa14 N=128 = pcgs (symmetric half storage),
# iiwa14 N=64 / N=16 = pcgc 2 rows per thread behind schur1, indy7 N=128 = pcgs, indy7 N=256 = the streaming pcg_kernel
SHAPES = [("indy7", 32), ("iiwa14", 128), ("iiwa14", 64), ("indy7", 128), ("indy7", 256), ("iiwa14", 16)]


def _ee(plant, q):
    from oracle import oracle as O
    return O.ee(plant, q)[0]


def _pair(plant, N, f64, ratio, iters=ITERS):
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    pr = mixed_problem(plant, N, ee=_ee)
    B = len(pr["kinds"])
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters, solve_ratio=ratio, **PCG)
    nat = NativeSolver(plant, N, B, f64=f64, dt=DT, **p)
    orc = OracleSolver(plant, N, B, dt=DT, f64=f64, **p)
    for s in (nat, orc):
        s.set_f_ext_batch(pr["f_ext"])
        s.set_cost_weights_batch(pr["w"])
    return nat, orc, pr


@pytest.mark.parametrize("plant,N", SHAPES)
@pytest.mark.parametrize("ratio", [0.3, 0.5, 1.0])
def test_convergence_and_solve_ratio_exit_fp32(plant, N, ratio):
    nat, orc, pr = _pair(plant, N, False, ratio)
    B, kinds = len(pr["kinds"]), pr["kinds"]
    rg = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    ro = orc.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    _check_record_semantics(ro, B, ratio, ITERS)
    _check_record_semantics(rg, B, ratio, ITERS)
    # a strict subset is converged at entry: E and U rows, nothing else
    entry = np.array([k in "EU" for k in kinds])
    np.testing.assert_array_equal(ro["pcg_iters_all"][0] == 0, entry)
    assert 0 < entry.sum() < B
    # HIP == oracle: counters, the truncated record's shape, and -- for every row whose fate is not fp32 chaos -- flags and the iteration
    # in which it converged.  (The fig-8 rows F are free-running fp32 iterates: after three or four iterations one of them takes a different
    # line-search branch somewhere -- DESIGN.md 3 -- so their records are compared over the first three iterations; the float64 pair below
    # compares ALL rows over ALL iterations exactly.)
    assert rg["iters_done"] == ro["iters_done"] and rg["ls_num_iters"] == ro["ls_num_iters"]
    np.testing.assert_array_equal(rg["sqp_iters"], ro["sqp_iters"])
    assert rg["pcg_iters"].shape == ro["pcg_iters"].shape
    fig8 = np.array([c == "F" for c in kinds])
    zg, zo = rg["pcg_iters_all"] == 0, ro["pcg_iters_all"] == 0
    np.testing.assert_array_equal(zg[:, ~fig8], zo[:, ~fig8])
    np.testing.assert_array_equal(zg[:3], zo[:3])
    np.testing.assert_array_equal(rg["kkt_converged"][~fig8], ro["kkt_converged"][~fig8])
    d = np.abs(rg["pcg_iters_all"].astype(int) - ro["pcg_iters_all"])
    assert np.all(d[:, ~fig8] <= 2), (rg["pcg_iters_all"], ro["pcg_iters_all"])
    assert np.all(d[:2] <= np.maximum(3, 0.15 * ro["pcg_iters_all"][:2])), (rg["pcg_iters_all"], ro["pcg_iters_all"])
    # line searches: the fig-8 rows take the oracle's steps in the first two searches (later ones are covered, without fp32 chaos, by
    # test_teacher_forced_iterations); a U row is converged at entry AND moved by a full step; rows at rest (E, U after its step, P) compare
    # merits that differ by rounding only, so they are compared by where they end up
    ls = ro["ls_num_iters"]
    if ls:
        k = min(ls, 2)
        np.testing.assert_array_equal(rg["ls_step_size"][:k][:, fig8], ro["ls_step_size"][:k][:, fig8])
        u = np.array([c == "U" for c in kinds])
        assert np.all(ro["ls_step_size"][0][u] == 1.0) and np.all(rg["ls_step_size"][0][u] == 1.0)
        assert np.all(np.abs(rg["XU"][u] - pr["xu"][u]).max(axis=1) > 0.1)
    assert np.abs(rg["XU"][~fig8] - ro["XU"][~fig8]).max() < 5e-5   # measured 1.3e-5 (a U row's full step of 0.5 N m through fp32 R^-1 r)
    if ratio == 0.3:
        assert ro["iters_done"] == 1 and ro["ls_num_iters"] == 0                       # exit in the first iteration: nothing moved
        np.testing.assert_array_equal(rg["XU"], pr["xu"])
    if ratio == 0.5:
        assert ro["iters_done"] >= 2 and ro["iters_done"] < ITERS                      # exit in a later iteration (the P rows have to converge first)
    if ratio == 1.0:
        assert ro["iters_done"] == ITERS and 0 < ro["kkt_converged"].sum() < B


@pytest.mark.parametrize("plant,N", SHAPES)
@pytest.mark.parametrize("ratio", [0.5, 1.0])
def test_convergence_and_solve_ratio_exit_float64(plant, N, ratio):
    """both implementations in double: every counter, flag, count and decision equal, iterates to 1e-9 (rows at rest: absolute 1e-9)"""
    nat, orc, pr = _pair(plant, N, True, ratio)
    B = len(pr["kinds"])
    rg = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    ro = orc.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    _check_record_semantics(rg, B, ratio, ITERS)
    assert rg["iters_done"] == ro["iters_done"] and rg["ls_num_iters"] == ro["ls_num_iters"]
    for k in ("sqp_iters", "kkt_converged", "pcg_iters", "pcg_iters_all"):
        np.testing.assert_array_equal(rg[k], ro[k], err_msg=k)
    np.testing.assert_array_equal(rg["ls_step_size"], ro["ls_step_size"])        # also of the rows at rest: same decisions on rounding-level merits
    assert np.abs(rg["XU"] - ro["XU"]).max() <= 1e-9 * np.abs(ro["XU"]).max()
    for k in ("final_merit", "ls_min_merit"):
        assert np.abs(rg[k] - ro[k]).max() <= 1e-9 * max(1.0, np.abs(ro[k]).max()), k


def test_early_exit_leaves_the_reference_records():
    """bsqp.cuh:139,165 vs bindings.cu:111-128 on the device: an exit in iteration i leaves i+1 PCG records and i line-search records; the
    binding's dict shows i of each; drho is back at its default (bsqp.cuh:189) and rho is not; a second solve starts clean."""
    nat, orc, pr = _pair("indy7", 32, False, 0.5)
    rg = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    ro = orc.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    it = rg["iters_done"]
    assert 2 <= it < ITERS and rg["ls_num_iters"] == it - 1 and rg["pcg_iters"].shape[0] == it - 1 and rg["pcg_iters_all"].shape[0] == it
    assert np.all(nat.read("drho") == 1.0) and np.array_equal(nat.read("drho"), orc.buf("drho"))
    live = np.array([c == "F" for c in pr["kinds"]])
    np.testing.assert_allclose(nat.read("rho")[live], orc.buf("rho")[live], rtol=1e-6)
    # the flags of one solve do not leak into the next (bsqp.cuh:112-114, 186-188)
    for s in (nat, orc):
        s.reset_dual(); s.reset_rho()
    r2, o2 = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"]), orc.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    np.testing.assert_array_equal(r2["XU"], rg["XU"])
    np.testing.assert_array_equal(r2["kkt_converged"], rg["kkt_converged"])
    assert r2["iters_done"] == it == o2["iters_done"]


def test_a_pcg_breakdown_is_carried_like_the_reference_carries_it():
    """fp32 PCG can break down (rho or p^T A p overflow / 0 : 0 -> a non-finite alpha): the reference has no guard (pcg.cuh:96-141), its lambda of that
    trajectory is NaN from then on, every later merit of it is NaN, every later line search compares false and rejects (line_search.cuh:39-63), and the
    trajectory keeps the iterate it had.  Found in round 5 on indy7 N = 128 (fig-8 rows 20 and 58 with a random wrench, second SQP iteration, both in
    the oracle and on the device): the HIP path must break down on the SAME rows, reject the same steps, and keep every returned number finite."""
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    from oracle.oracle import OracleSolver
    plant, N, B = "indy7", 128, 64
    pr = fig8_problem(plant, N, B, f_ext_std=1.0)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6)
    nat = NativeSolver(plant, N, B, dt=0.01, **p)
    orc = OracleSolver(plant, N, B, dt=0.01, threads=os.cpu_count() or 1, **p)
    for s in (nat, orc):
        s.set_f_ext_batch(pr["f_ext"])
    rg = nat.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    ro = orc.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    bad_g = ~np.isfinite(nat.read("lambda").reshape(B, N + 2, nat.nx)[:, 1:N + 1]).all(axis=(1, 2))
    bad_o = ~np.isfinite(orc.buf("lambda")[:, 1:N + 1]).all(axis=(1, 2))
    assert bad_o.sum() >= 1, "the configuration no longer breaks down in the oracle: pick another one"
    np.testing.assert_array_equal(bad_g, bad_o)
    np.testing.assert_array_equal(rg["ls_step_size"][:, bad_o], ro["ls_step_size"][:, bad_o])
    assert np.all(rg["ls_step_size"][-1, bad_o] == -1.0)                                # rejected ever since
    for k in ("XU", "final_merit", "initial_merit", "ls_min_merit"):
        assert np.all(np.isfinite(rg[k])), k                                             # what the caller sees stays finite: the trajectory stands still
    np.testing.assert_array_equal(rg["pcg_iters_all"][:, bad_o] >= 200, ro["pcg_iters_all"][:, bad_o] >= 200)
