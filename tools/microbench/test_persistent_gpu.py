"""The persistent form of the SQP loop (kernels.hpp: sqp_pair_kernel -- ONE launch per solve, two trajectories per workgroup, no batch-wide
synchronisation) against the launched loop (three launches per SQP iteration, solver.hip:enqueue_solve): the same device functions, so the
SAME BITS -- iterates, duals, rho, merits, every per-iteration record -- on figure-8 batches of every parity of size and on the mixed batch
of tests/mixed_batch.py, where the exit rule of bsqp.cuh:142-167 acts (a workgroup whose trajectories are all converged waits for the whole
batch's count; a workgroup with an unconverged one never asks)."""
import os

import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem

pytestmark = pytest.mark.gpu
DT = 0.01
KEYS = ("XU", "final_merit", "initial_merit", "pcg_iters_all", "pcg_iters", "ls_step_size", "ls_min_merit", "kkt_converged", "sqp_iters")


def _solver(persist, N, B, **p):
    from gato_amd._lib import NativeSolver
    old = os.environ.get("GATO_PERSIST")
    os.environ["GATO_PERSIST"] = "1" if persist else "0"      # read once, when the solver is created (solver.hip:plan_pcg)
    try:
        return NativeSolver("indy7", N, B, dt=DT, **p)
    finally:
        if old is None:
            del os.environ["GATO_PERSIST"]
        else:
            os.environ["GATO_PERSIST"] = old


def _same(a, b, sa, sb):
    assert a["iters_done"] == b["iters_done"] and a["ls_num_iters"] == b["ls_num_iters"]
    for k in KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for name in ("rho", "drho", "lambda", "dz", "merit_cur"):
        np.testing.assert_array_equal(sa.read(name), sb.read(name), err_msg=name)


@pytest.mark.parametrize("B,iters,fstd", [(1, 3, 0.0), (2, 4, 2.0), (5, 3, 0.0), (64, 10, 0.0), (1024, 10, 0.0)])
def test_persistent_loop_equals_the_launched_loop(B, iters, fstd):
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters)
    pr = fig8_problem("indy7", 32, B, f_ext_std=fstd)
    out, sv = [], []
    for persist in (False, True):
        s = _solver(persist, 32, B, **p)
        s.set_f_ext_batch(pr["f_ext"])
        r = s.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
        r2 = s.solve(r["XU"], DT, pr["x_s"], pr["ref"])      # a second, warm-started solve on the same handle (lambda and rho persist)
        out.append((r, r2)); sv.append(s)
    _same(out[0][0], out[1][0], sv[0], sv[1]) if False else None
    for k in KEYS:
        np.testing.assert_array_equal(out[0][0][k], out[1][0][k], err_msg="first solve: " + k)
    _same(out[0][1], out[1][1], sv[0], sv[1])
    assert out[1][0]["iters_done"] == iters and np.all(np.isfinite(out[1][0]["XU"]))


@pytest.mark.parametrize("kinds,iters", [("EUPPFFEUPPFF", 6), ("EUEU", 4), ("PPEUPPE", 8), ("EU", 3), ("E", 2), ("PFPF", 6)])
def test_persistent_loop_on_the_mixed_batch(kinds, iters):
    """strict subsets converged (no exit at solve_ratio 1), everything converged at entry (exit in the first iteration: nothing moves, one PCG
    record more than line searches), everything converging later (exit in a later iteration), odd batches (a workgroup with one trajectory)"""
    from mixed_batch import check_record_semantics, mixed_problem
    from oracle import oracle as O
    N, B = 32, len(kinds)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=iters, solve_ratio=1.0, pcg_tol=1e-8, max_pcg_iters=1000)
    pr = mixed_problem("indy7", N, kinds=kinds, ee=lambda pl, q: O.ee(pl, q)[0])
    out, sv = [], []
    for persist in (False, True):
        s = _solver(persist, N, B, **p)
        s.set_f_ext_batch(pr["f_ext"]); s.set_cost_weights_batch(pr["w"])
        out.append(s.solve(pr["xu"], DT, pr["x_s"], pr["ref"])); sv.append(s)
    _same(out[0], out[1], sv[0], sv[1])
    check_record_semantics(out[1], B, 1.0, iters)
    if "F" not in kinds:
        assert out[1]["iters_done"] < iters, "every row converges: the rule must fire"


def test_persistent_loop_is_not_used_where_it_would_be_wrong():
    """solve_ratio < 1 (the rule can fire while a workgroup still holds an unconverged trajectory) and sharded handles take the launched loop;
    the results say so by being those of the launched loop"""
    from mixed_batch import mixed_problem
    from oracle import oracle as O
    kinds, N = "EUPPFFEUPPFF", 32
    B = len(kinds)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6, solve_ratio=0.5, pcg_tol=1e-8, max_pcg_iters=1000)
    pr = mixed_problem("indy7", N, kinds=kinds, ee=lambda pl, q: O.ee(pl, q)[0])
    out, sv = [], []
    for persist in (False, True):
        s = _solver(persist, N, B, **p)
        s.set_f_ext_batch(pr["f_ext"]); s.set_cost_weights_batch(pr["w"])
        out.append(s.solve(pr["xu"], DT, pr["x_s"], pr["ref"])); sv.append(s)
    _same(out[0], out[1], sv[0], sv[1])
    assert 2 <= out[1]["iters_done"] < 6
