"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, solve their rows with the ORACLE standing in for the per-rank
solver (this test is about the sharding/gather layer, not the kernels) and all_gather; the result equals the unsharded solve
bit for bit per trajectory."""
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


def _worker(rank, world, port, N, B, q):
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
    pr = fig8_problem("indy7", N, hi - lo, batch_offset=lo)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2)
    check_sharded_params(p["solve_ratio"], world)
    s = OracleSolver("indy7", N, hi - lo, dt=0.01, **p)
    out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    # the per-solve data path of bench.py: one packed buffer per rank, ONE collective
    pk = PackedResults(hi - lo, out["XU"].shape[1], world)
    pk.xu.copy_(torch.from_numpy(out["XU"]))
    pk.merit.copy_(torch.from_numpy(out["final_merit"]))
    pk.all_gather()
    g = {"XU": pk.global_xu().numpy().copy(), "final_merit": pk.global_merit().numpy().copy()}
    g.update(gather_results({"sqp_iters": out["sqp_iters"]}))   # statistics go through the generic gather
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


def test_shard_bounds():
    from gato_amd.sharding import check_sharded_params, shard_bounds
    assert shard_bounds(8192, 8, 3) == (3072, 4096)
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)
    check_sharded_params(1.0, 8)
    check_sharded_params(0.5, 1)          # a single rank counts over the whole batch: fine
    with pytest.raises(ValueError):
        check_sharded_params(0.5, 2)      # the solved count couples the shards (bsqp.cuh:165): refused, not approximated
