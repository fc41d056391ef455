"""The N > 1 path on real devices: two ranks over RCCL ("nccl"), each with its own NativeSolver on its own GPU, the packed
one-collective gather of gato_amd/sharding.py -- the sharded result equals the single-GPU result bit for bit.  Needs two GPUs
(skipped on the 1-GPU box; the driver's scaling run exercises the same code through bench.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, B, q, one_device=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0 if one_device else rank)
    dev = torch.device("cuda", 0 if one_device else rank)
    if one_device:   # two ranks sharing the box's only GPU: RCCL refuses duplicate devices, gloo stages the collective through the host
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    from gato_amd.sharding import PackedResults, check_sharded_params, shard_bounds
    lo, hi = shard_bounds(B, world, rank)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3)
    check_sharded_params(p["solve_ratio"], world)
    pr = fig8_problem("indy7", N, hi - lo, batch_offset=lo)
    s = NativeSolver("indy7", N, hi - lo, dt=0.01, **p)          # bound to cuda:rank (the device current at creation)
    pk = PackedResults(hi - lo, s.traj, world, dev)
    pk.xu.copy_(torch.from_numpy(pr["xu"]).to(dev))
    xs, ref = torch.from_numpy(pr["x_s"]).to(dev), torch.from_numpy(pr["ref"]).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    s.solve_device(pk.xu.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st)
    s.copy_final_merit_device(pk.merit.data_ptr(), st)
    if one_device:
        torch.cuda.synchronize()
        host = PackedResults(hi - lo, s.traj, world, "cpu")   # the same packed layout, gathered by the same single collective
        host.local.copy_(pk.local.cpu())
        host.all_gather()
        pk = host
    else:
        pk.all_gather()
        torch.cuda.synchronize()
    if rank == 0:
        q.put((pk.global_xu().cpu().numpy(), pk.global_merit().cpu().numpy(), pk.best()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_gpu_sharded_solve_equals_single_gpu():
    import torch.multiprocessing as mp
    N, B, world = 32, 64, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    xu, merit, best = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    pr = fig8_problem("indy7", N, B)
    one = NativeSolver("indy7", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3))
    ref = one.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    np.testing.assert_array_equal(xu, ref["XU"])
    np.testing.assert_array_equal(merit, ref["final_merit"])
    assert best[1] == int(np.argmin(ref["final_merit"]))


def test_two_ranks_on_one_gpu_sharded_solve_equals_single_batch():
    """The sharded PRODUCT path on the 1-GPU box: two processes, each with its own NativeSolver on cuda:0 solving its half of the batch
    (per-rank lambda / rho / order state, problem rows by batch_offset), the packed one-collective gather (gloo, staged through the host:
    RCCL will not take two ranks on one device) -- equals the single-process solve of the whole batch bit for bit."""
    import torch.multiprocessing as mp
    N, B, world = 32, 48, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    xu, merit, best = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    pr = fig8_problem("indy7", N, B)
    one = NativeSolver("indy7", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3))
    ref = one.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    np.testing.assert_array_equal(xu, ref["XU"])
    np.testing.assert_array_equal(merit, ref["final_merit"])
    assert best[1] == int(np.argmin(ref["final_merit"]))
