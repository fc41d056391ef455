"""The N > 1 path on real devices: two ranks, each with its own NativeSolver on its own GPU and the library's own RCCL communicator
(gato_comm_init: the solved count shared per SQP iteration, the packed one-collective gather) -- the sharded result equals the single-GPU
result bit for bit.  The two-GPU test is skipped on the 1-GPU box (the driver's scaling run exercises the same code through bench.py); what
the box CAN run is here too: a one-rank communicator through the same entry points, two ranks sharing the device with the solved counts
handed over by the test, and the sharded exit rule on a mixed batch against the unsharded solve."""
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
    from gato_amd.sharding import PackedResults, check_sharded_params, connect, shard_bounds
    lo, hi = shard_bounds(B, world, rank)
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3)
    pr = fig8_problem("indy7", N, hi - lo, batch_offset=lo)
    s = NativeSolver("indy7", N, hi - lo, dt=0.01, **p)          # bound to cuda:rank (the device current at creation)
    if one_device:   # no RCCL between two ranks on one device: the other shard's solved counts (none converge here) are handed over
        s.debug_set_remote_solved(np.zeros(3, np.uint32), B)
    else:
        connect(s)   # the library's own communicator: ncclAllReduce of the solved count per SQP iteration, ncclAllGather of the results
    check_sharded_params(p["solve_ratio"], world, coupled=True)
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
        pk.all_gather(solver=s, stream=st)
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


def test_one_rank_communicator_through_the_same_entry_points():
    """gato_comm_unique_id / gato_comm_init / the per-iteration ncclAllReduce / gato_gather_results with world size 1 -- everything the 8-GPU run
    calls, on the box's one device: a solver with the communicator gives the bits of one without, the gather returns the packed buffer."""
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    from gato_amd.sharding import PackedResults
    N, B = 32, 24
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=3)
    pr = fig8_problem("indy7", N, B)
    plain = NativeSolver("indy7", N, B, dt=0.01, **p).solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    s = NativeSolver("indy7", N, B, dt=0.01, **p)
    uid = NativeSolver.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    s.comm_init(uid, 1, 0)
    r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    for k in ("XU", "final_merit", "pcg_iters_all", "ls_step_size", "kkt_converged"):
        np.testing.assert_array_equal(r[k], plain[k], err_msg=k)
    graph = os.environ.get("GATO_GRAPH", "0") not in ("", "0")       # a captured solve cannot take the host look: it shares the count per iteration
    assert s.shard_stats() == {"deferred_solves": 0 if graph else 1, "replays": 0}   # the default: ONE ncclAllReduce of the count vector behind the solve
    s.set_solved_count_mode("per_iteration")                          # round 3's form: a 4-byte ncclAllReduce in every SQP iteration
    s.reset_dual(); s.reset_rho()
    r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    for k in ("XU", "final_merit", "pcg_iters_all", "ls_step_size", "kkt_converged"):
        np.testing.assert_array_equal(r[k], plain[k], err_msg=k)
    assert s.shard_stats() == {"deferred_solves": 0 if graph else 1, "replays": 0}
    dev = torch.device("cuda", 0)
    pk = PackedResults(B, s.traj, 1, dev)
    pk.xu.copy_(torch.from_numpy(r["XU"]).to(dev))
    out = torch.zeros_like(pk.local)
    s.gather_results(pk.local.data_ptr(), out.data_ptr(), pk.n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, pk.local)
    s.comm_destroy()
    r2 = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])     # back to the unsharded rule; warm-started duals: a different solve, still finite
    assert np.all(np.isfinite(r2["XU"]))
    with pytest.raises(Exception):
        s.gather_results(pk.local.data_ptr(), out.data_ptr(), pk.n, 0)   # no communicator any more


@pytest.mark.parametrize("mode", ["deferred", "per_iteration", "graph", "graph_recapture"])
@pytest.mark.parametrize("ratio", [0.5, 1.0])
def test_sharded_exit_rule_against_the_unsharded_solve(ratio, mode):
    """bsqp.cuh:165 on a sharded batch, product path: the mixed batch of tests/mixed_batch.py cut in two shards (the first holds every early
    converger, the second none), each shard a NativeSolver that counts its own rows and is GIVEN the other shard's solved count per SQP
    iteration (what the all-reduce delivers on two GPUs): both shards exit in the whole batch's iteration and reproduce their rows of the
    unsharded solve bit for bit -- also with solve_ratio 1, where the converged shard must keep stepping its rows.
    mode: how the count reaches the rule --
      deferred        (round 4, the default) speculative solve + ONE reduction of the count vector; ratio 0.5 fires the rule: the REPLAY path
                      (snapshot restored, exact re-run); ratio 1 never reaches the threshold: the NO-REPLAY path
      per_iteration   one reduction per SQP iteration (round 3)
      graph           hipGraph replay of the host-buffer solve captured while sharded (per-iteration reduction nodes inside the graph)
      graph_recapture a graph captured UNSHARDED must not survive the switch to the sharded rule (advisor, round 3: the cache key ignored it)"""
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from mixed_batch import mixed_problem
    from oracle import oracle as O
    N, kinds = 32, "EUPPEUFFFFFF"
    B, H = len(kinds), 6
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6, solve_ratio=ratio, pcg_tol=1e-8, max_pcg_iters=1000)
    ee = lambda pl, q: O.ee(pl, q)[0]  # noqa: E731
    pr = mixed_problem("indy7", N, kinds=kinds, ee=ee)
    one = NativeSolver("indy7", N, B, dt=0.01, **p)
    one.set_f_ext_batch(pr["f_ext"]); one.set_cost_weights_batch(pr["w"])
    ref = one.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    solved = np.cumsum(ref["pcg_iters_all"] == 0, axis=0) > 0          # [iteration, row]: counted as solved in that iteration
    for lo, hi in ((0, H), (H, B)):
        other = np.r_[0:lo, hi:B]
        sh = mixed_problem("indy7", N, kinds=kinds, ee=ee, rows=(lo, hi))
        s = NativeSolver("indy7", N, hi - lo, dt=0.01, **p)
        s.set_f_ext_batch(sh["f_ext"]); s.set_cost_weights_batch(sh["w"])
        if mode == "graph_recapture":
            s.set_graph_mode(True)
            first = s.solve(sh["xu"], 0.01, sh["x_s"], sh["ref"])      # captured with the UNSHARDED rule (a different solve: the shard alone)
            assert np.all(np.isfinite(first["XU"]))
            s.reset_dual(); s.reset_rho()
        s.debug_set_remote_solved(solved[:, other].sum(axis=1).astype(np.uint32), B)
        if mode == "per_iteration":
            s.set_solved_count_mode("per_iteration")
        if mode == "graph":
            s.set_graph_mode(True)
        r = s.solve(sh["xu"], 0.01, sh["x_s"], sh["ref"])
        assert r["iters_done"] == ref["iters_done"] and r["ls_num_iters"] == ref["ls_num_iters"]
        for k in ("XU", "final_merit", "kkt_converged", "sqp_iters"):
            np.testing.assert_array_equal(r[k], ref[k][lo:hi], err_msg=k)
        np.testing.assert_array_equal(r["pcg_iters_all"], ref["pcg_iters_all"][:, lo:hi])
        np.testing.assert_array_equal(r["ls_step_size"], ref["ls_step_size"][:, lo:hi])
        np.testing.assert_array_equal(s.read("rho"), one.read("rho")[lo:hi])
        np.testing.assert_array_equal(s.read("lambda").reshape(hi - lo, -1), one.read("lambda").reshape(B, -1)[lo:hi])
        st = s.shard_stats()
        if mode == "deferred" and os.environ.get("GATO_GRAPH", "0") in ("", "0"):   # (a suite run under GATO_GRAPH=1 captures every solve: per-iteration counts)
            # the rule fires at ratio 0.5 (replay) and never at ratio 1 (the speculative run IS the result)
            assert st == {"deferred_solves": 1, "replays": 1 if ratio == 0.5 else 0}, st
        else:
            assert st["deferred_solves"] == 0, st
    if ratio == 0.5:
        assert 2 <= ref["iters_done"] < 6
    else:
        assert ref["iters_done"] == 6 and solved[4, :H].all() and not solved[5, H:].all()   # per-shard counting would have stopped shard 0 early


def test_replay_backs_off_to_per_iteration_counts_and_a_captured_solve_never_waits():
    """Advisor (round 4): a batch whose exit rule fires would pay a speculative pass plus an exact replay on EVERY solve, and the deferred form's
    host wait inside gato_solve_device breaks a caller's stream capture.  Now: after a replay the next 8 sharded solves count per iteration
    (same bits, one pass each), and a solve enqueued on a stream that is being captured takes the per-iteration form by itself."""
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from mixed_batch import mixed_problem
    from oracle import oracle as O
    N, kinds = 32, "EUPPEUFFFFFF"
    B, H = len(kinds), 6
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6, solve_ratio=0.5, pcg_tol=1e-8, max_pcg_iters=1000)
    ee = lambda pl, q: O.ee(pl, q)[0]  # noqa: E731
    pr = mixed_problem("indy7", N, kinds=kinds, ee=ee)
    one = NativeSolver("indy7", N, B, dt=0.01, **p)
    one.set_f_ext_batch(pr["f_ext"]); one.set_cost_weights_batch(pr["w"])
    ref = one.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    solved = np.cumsum(ref["pcg_iters_all"] == 0, axis=0) > 0
    lo, hi = 0, H
    sh = mixed_problem("indy7", N, kinds=kinds, ee=ee, rows=(lo, hi))
    s = NativeSolver("indy7", N, hi - lo, dt=0.01, **p)
    s.set_f_ext_batch(sh["f_ext"]); s.set_cost_weights_batch(sh["w"])
    s.debug_set_remote_solved(solved[:, H:].sum(axis=1).astype(np.uint32), B)
    graph = os.environ.get("GATO_GRAPH", "0") not in ("", "0")
    for n in range(10):
        s.reset_dual(); s.reset_rho()
        r = s.solve(sh["xu"], 0.01, sh["x_s"], sh["ref"])
        np.testing.assert_array_equal(r["XU"], ref["XU"][lo:hi])
        np.testing.assert_array_equal(r["pcg_iters_all"], ref["pcg_iters_all"][:, lo:hi])
        if not graph:
            # solve 0: speculative + replay; solves 1..8: per iteration (the back-off); solve 9: speculative again (+ replay, back-off now 16)
            assert s.shard_stats() == {"deferred_solves": 1 if n < 9 else 2, "replays": 1 if n < 9 else 2}, (n, s.shard_stats())
    # a caller's capture of gato_solve_device on a sharded handle: no host wait inside, the graph replays to the same bits
    s2 = NativeSolver("indy7", N, hi - lo, dt=0.01, **dict(p, solve_ratio=1.0))
    s2.set_f_ext_batch(sh["f_ext"]); s2.set_cost_weights_batch(sh["w"])
    s2.debug_set_remote_solved(np.zeros(6, np.uint32), B)
    eager = s2.solve(sh["xu"], 0.01, sh["x_s"], sh["ref"])
    before = s2.shard_stats()["deferred_solves"]
    dev = torch.device("cuda", 0)
    xu0 = torch.from_numpy(sh["xu"]).to(dev)
    xu, xs, rf = xu0.clone(), torch.from_numpy(sh["x_s"]).to(dev), torch.from_numpy(sh["ref"]).to(dev)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    s2.reset_dual(); s2.reset_rho()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        s2.reset_async(True, True, side.cuda_stream)
        s2.solve_device(xu.data_ptr(), 0.01, xs.data_ptr(), rf.data_ptr(), side.cuda_stream)
    assert s2.shard_stats()["deferred_solves"] == before                 # the capture took the per-iteration form
    xu.copy_(xu0)
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(xu.cpu().numpy(), eager["XU"])


@pytest.mark.parametrize("ratio", [0.5, 1.0], ids=["replay", "no-replay"])
def test_the_verdict_of_a_deferred_solve_is_taken_by_the_next_entry_point(ratio):
    """Round 6: gato_solve_device on a sharded handle returns with the verdict PENDING -- the host is not held for the whole solve -- and the next entry
    point that reads or changes what the solve reads or writes takes it (and replays exactly if the exit rule fired).  The device-pointer sequence of
    bench.py's loop (reset_async, solve_device, a gather that does NOT take the verdict, copy_final_merit_device that does) on the shard that holds the
    early convergers of the mixed batch: nothing is counted as settled before the merit copy, the results are the unsharded solve's bits afterwards --
    through the replay at ratio 0.5, without one at ratio 1 -- and a second solve enqueued straight behind it (reset_async settles first) repeats them."""
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from mixed_batch import mixed_problem
    from oracle import oracle as O
    if os.environ.get("GATO_GRAPH", "0") not in ("", "0"):
        pytest.skip("a suite run under GATO_GRAPH=1 is about the host-buffer solve")
    N, kinds = 32, "EUPPEUFFFFFF"
    B, H = len(kinds), 6
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=6, solve_ratio=ratio, pcg_tol=1e-8, max_pcg_iters=1000)
    ee = lambda pl, q: O.ee(pl, q)[0]  # noqa: E731
    pr = mixed_problem("indy7", N, kinds=kinds, ee=ee)
    one = NativeSolver("indy7", N, B, dt=0.01, **p)
    one.set_f_ext_batch(pr["f_ext"]); one.set_cost_weights_batch(pr["w"])
    ref = one.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    solved = np.cumsum(ref["pcg_iters_all"] == 0, axis=0) > 0
    sh = mixed_problem("indy7", N, kinds=kinds, ee=ee, rows=(0, H))
    s = NativeSolver("indy7", N, H, dt=0.01, **p)
    s.set_f_ext_batch(sh["f_ext"]); s.set_cost_weights_batch(sh["w"])
    s.debug_set_remote_solved(solved[:, H:].sum(axis=1).astype(np.uint32), B)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    xu0 = torch.from_numpy(sh["xu"]).to(dev)
    xs, rf = torch.from_numpy(sh["x_s"]).to(dev), torch.from_numpy(sh["ref"]).to(dev)
    merit = torch.zeros(H, device=dev)
    for rep in range(2):
        xu = xu0.clone()
        s.reset_async(True, True, st)            # (second pass: takes nothing -- the first pass's verdict was taken by its merit copy)
        before = s.shard_stats()
        s.solve_device(xu.data_ptr(), 0.01, xs.data_ptr(), rf.data_ptr(), st)
        assert s.shard_stats() == before         # enqueued, not settled: neither counted as a speculative solve nor replayed yet
        s.copy_final_merit_device(merit.data_ptr(), st)   # the next entry point on the handle: the verdict, the replay if the rule fired, then the copy
        after = s.shard_stats()
        assert after["deferred_solves"] == before["deferred_solves"] + 1 and after["replays"] == before["replays"] + (1 if ratio == 0.5 else 0), (before, after)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(xu.cpu().numpy(), ref["XU"][:H])
        np.testing.assert_array_equal(merit.cpu().numpy(), ref["final_merit"][:H])
        r = s.stats()
        np.testing.assert_array_equal(r["pcg_iters_all"], ref["pcg_iters_all"][:, :H])
        assert r["iters_done"] == ref["iters_done"]
        if ratio == 0.5:
            break    # (after a replay the next 8 solves count per iteration: test_replay_backs_off_...; the second pass is about the no-replay path)
    # the getters take a pending verdict too: solve_device straight into stats()
    if ratio == 1.0:
        xu = xu0.clone()
        s.reset_async(True, True, st)
        s.solve_device(xu.data_ptr(), 0.01, xs.data_ptr(), rf.data_ptr(), st)
        r = s.stats()                            # gato_get_*: sync_last -> settle -> drain
        np.testing.assert_array_equal(xu.cpu().numpy(), ref["XU"][:H])
        np.testing.assert_array_equal(r["final_merit"], ref["final_merit"][:H])
