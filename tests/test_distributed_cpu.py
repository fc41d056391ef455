"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, solve their rows with the ORACLE standing in for the per-rank
solver (this test is about the sharding / gather / solved-count layer, not the kernels) and all_gather; the result equals the unsharded solve
bit for bit per trajectory -- also where the exit rule acts: a mixed batch with a strict subset converged (tests/mixed_batch.py), solve_ratio
below and at 1, where per-shard counting would stop one shard in a different iteration than the whole batch."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _share_solved_count(solver, global_batch):
    """the 4-byte SUM all-reduce per SQP iteration for the oracle standing in for a rank's solver"""
    import torch
    import torch.distributed as dist

    def reduce(n, it):
        t = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(t)
        return int(t.item())
    solver.set_shard(reduce, global_batch)


def _worker(rank, world, port, N, B, q, mixed=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    import torch
    from gato_amd.sharding import PackedResults, check_sharded_params, gather_results, shard_bounds
    from oracle.oracle import OracleSolver
    lo, hi = shard_bounds(B, world, rank)
    if mixed is None:
        pr = fig8_problem("indy7", N, hi - lo, batch_offset=lo)
        p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2)
    else:
        from mixed_batch import mixed_problem
        from oracle import oracle as O
        pr = mixed_problem("indy7", N, kinds=mixed["kinds"], ee=lambda pl, qq: O.ee(pl, qq)[0], rows=(lo, hi))
        p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6, solve_ratio=mixed["ratio"], pcg_tol=1e-8, max_pcg_iters=1000)
    s = OracleSolver("indy7", N, hi - lo, dt=0.01, **p)
    if mixed is not None:
        s.set_f_ext_batch(pr["f_ext"]); s.set_cost_weights_batch(pr["w"])
    _share_solved_count(s, B)
    check_sharded_params(p["solve_ratio"], world, coupled=True)
    out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    # the per-solve data path of bench.py: one packed buffer per rank, ONE collective
    pk = PackedResults(hi - lo, out["XU"].shape[1], world)
    pk.xu.copy_(torch.from_numpy(out["XU"]))
    pk.merit.copy_(torch.from_numpy(out["final_merit"]))
    pk.all_gather()
    g = {"XU": pk.global_xu().numpy().copy(), "final_merit": pk.global_merit().numpy().copy()}
    g.update(gather_results({"sqp_iters": out["sqp_iters"], "kkt_converged": out["kkt_converged"]}))   # statistics go through the generic gather
    g["iters_done"], g["ls_num_iters"] = out["iters_done"], out["ls_num_iters"]
    best = pk.best()
    dist.barrier()
    if rank == 0:
        q.put((g, best))
    dist.destroy_process_group()


def test_sharded_solve_equals_unsharded():
    N, B, world = 8, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    g, best = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    from oracle.oracle import OracleSolver
    pr = fig8_problem("indy7", N, B)
    s = OracleSolver("indy7", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2))
    ref = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    np.testing.assert_array_equal(g["XU"], ref["XU"])
    np.testing.assert_array_equal(g["final_merit"], ref["final_merit"])
    assert g["XU"].shape == (B, 18 * N - 6) and np.all(g["sqp_iters"] == ref["sqp_iters"])
    assert best[1] == int(np.argmin(ref["final_merit"])) and abs(best[0] - float(ref["final_merit"].min())) < 1e-6


@pytest.mark.parametrize("ratio", [0.5, 1.0])
def test_sharded_exit_rule_on_a_mixed_batch(ratio):
    """The solved count shared per SQP iteration: the shard that holds the early convergers (E U P P rows first) and the shard that holds
    none of them exit in the iteration the WHOLE batch exits in (ratio 0.5), and with ratio 1 the converged shard's rows keep being
    stepped until the end -- bit for bit the unsharded solve.  (Counting per shard, rank 0 would stop iterations earlier.)"""
    N, world = 8, 2
    kinds = "EUPPEUFFFFFF"      # rank 0: E U P P E U (all converge within a few iterations), rank 1: six fig-8 rows
    B = len(kinds)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, q, dict(kinds=kinds, ratio=ratio))) for r in range(world)]
    for p in procs:
        p.start()
    g, _ = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from mixed_batch import mixed_problem
    from oracle import oracle as O
    from oracle.oracle import OracleSolver
    pr = mixed_problem("indy7", N, kinds=kinds, ee=lambda pl, qq: O.ee(pl, qq)[0])
    s = OracleSolver("indy7", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6, solve_ratio=ratio, pcg_tol=1e-8, max_pcg_iters=1000))
    s.set_f_ext_batch(pr["f_ext"]); s.set_cost_weights_batch(pr["w"])
    ref = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    assert g["iters_done"] == ref["iters_done"] and g["ls_num_iters"] == ref["ls_num_iters"]
    np.testing.assert_array_equal(g["XU"], ref["XU"])
    np.testing.assert_array_equal(g["final_merit"], ref["final_merit"])
    np.testing.assert_array_equal(g["kkt_converged"], ref["kkt_converged"])
    np.testing.assert_array_equal(g["sqp_iters"], ref["sqp_iters"])
    # the cases are not vacuous
    first = np.where((ref["pcg_iters_all"] == 0).any(axis=0), (ref["pcg_iters_all"] == 0).argmax(axis=0), 99)
    if ratio == 1.0:   # rank 0's rows are ALL converged before the last iteration (per-shard counting would have stopped it there), rank 1's are not
        assert ref["iters_done"] == 6 and first[:6].max() < 5 and not ref["kkt_converged"][6:].all()
    else:              # the whole batch exits in a later iteration, before its line search; rank 0 alone (4 of its 6 rows converged at entry) would have left in the first
        assert 2 <= ref["iters_done"] < 6 and ref["ls_num_iters"] == ref["iters_done"] - 1 and (first[:6] == 0).sum() >= 3


def test_shard_bounds():
    from gato_amd.sharding import check_sharded_params, shard_bounds
    assert shard_bounds(8192, 8, 3) == (3072, 4096)
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)
    check_sharded_params(0.5, 1)                  # a single rank counts over the whole batch: fine
    check_sharded_params(0.5, 2, coupled=True)    # the ranks share the solved count: exact for any solve_ratio
    with pytest.raises(ValueError):
        check_sharded_params(0.5, 2)              # uncoupled shards: refused, not approximated ...
    with pytest.raises(ValueError):
        check_sharded_params(1.0, 8)              # ... also at solve_ratio 1 (a converged shard would stop stepping its rows early)


class _FakeSolver:
    """what sharding.connect needs of a solver, with a scripted failure: the protocol between the ranks is the thing under test"""

    def __init__(self, rank, fail_at):
        self.rank, self.fail_at, self.calls = rank, fail_at, []

    def comm_available(self):
        self.calls.append("available")
        return "librccl.so missing (scripted)" if self.fail_at == "available" else None

    def comm_unique_id(self):
        self.calls.append("unique_id")
        return bytes(range(128))

    def comm_init_rank(self, uid, world, rank):
        self.calls.append("init_rank")
        assert uid == bytes(range(128)) and rank == self.rank
        if self.fail_at == "init_rank":
            raise RuntimeError("ncclCommInitRank: unhandled system error (scripted, rank %d)" % rank)

    def comm_confirm(self):
        self.calls.append("confirm")
        if self.fail_at == "confirm":
            raise RuntimeError("hipErrorOutOfMemory (scripted)")

    def comm_destroy(self):
        self.calls.append("destroy")


def _connect_worker(rank, world, port, fail_rank, fail_at, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gato_amd.sharding import connect
    s = _FakeSolver(rank, fail_at if rank == fail_rank else None)
    try:
        connect(s, timeout_s=30.0)
        out = "ok"
    except RuntimeError as e:
        out = str(e)
    q.put((rank, out, s.calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_at", [None, "available", "init_rank", "confirm"])
def test_connect_fails_on_every_rank_or_on_none(fail_at):
    """sharding.connect: whatever step fails on ONE rank (RCCL missing, ncclCommInitRank after the probe passed, the count-mode agreement), BOTH
    ranks raise with that rank and its reason, the step after it is never entered on any rank (no rank is left inside a collective on a
    communicator whose peer is gone), and every rank that had a communicator dropped it."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_connect_worker, args=(r, world, port, 1, fail_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, out, calls = q.get(timeout=120)
        got[r] = (out, calls)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        out, calls = got[r]
        if fail_at is None:
            assert out == "ok" and calls == ["available"] + (["unique_id"] if r == 0 else []) + ["init_rank", "confirm"]
            continue
        assert out != "ok" and "1:" in out.replace("{", "").replace(" ", "") and "scripted" in out, out     # the failing rank and ITS reason, on both ranks
        if fail_at == "available":
            assert "init_rank" not in calls and "destroy" not in calls
        elif fail_at == "init_rank":
            assert "confirm" not in calls and calls[-1] == "destroy"      # nobody issued a collective on the half-built communicator
        else:
            assert calls[-2:] == ["confirm", "destroy"]
