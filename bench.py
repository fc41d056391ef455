#!/usr/bin/env python3
"""Headline benchmark: SQP iterations/sec (whole node), indy7 N=32 batch=1024 per GPU, figure-8 tracking (BASELINE.json, config C2/C4).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one batch: reset_dual() + reset_rho() + a full `BSQP::solve` (max_sqp_iters = 10, solve_ratio = 1,
DEFAULT_SOLVER_PARAMS otherwise) on B = 1024 trajectories per GPU, inputs already resident in HBM, plus -- for N > 1 -- the
solved count of the exit rule (deferred: ONE ncclAllReduce of the per-iteration count vector behind a speculative solve, an exact replay if the
rule fired; GATO_SOLVED_COUNT=periter: a 4-byte ncclAllReduce in every SQP iteration) and the all-gather of iterates and merits over xGMI.
The library leaves the verdict of a sharded solve to its NEXT entry point (gato_abi.h): the gather of solve n is enqueued between the launches of solve
n + 1 and that solve's merit copy (which takes the verdict), so it runs on the communication stream beside solve n + 1 and the host never holds the device idle.
value = sum over ranks of B * iterations / wall time (max over ranks).  Rank 0 prints ONE JSON line.

    --workload hparam --plant iiwa14 --knots 64 --batch 512     BASELINE config C5 (the hyper-parameter sweep): rank g solves shard g =
                                                                cost tuple g of the notebook's grid, per-trajectory rho, dt 0.05, mu 1, pcg_tol 1e-3
    --rehearse-one-device    N > 1 ranks on a box with ONE GPU (a rehearsal of the world > 1 branch below, never a measurement): every rank
                             uses cuda:0, torch.distributed runs on gloo, RCCL refuses the duplicate device inside gato_comm_init, so the
                             verified fallback takes over (results through torch.distributed, the other shards' solved counts handed to
                             the library as zeros) -- every line of the multi-rank path runs except RCCL's own kernels
    --as-rank R --of G       ONE process, ONE device, but the rows of rank R of a G-rank job (fig-8: rows [R B, (R+1) B) of the global batch; the sweep:
                             shard R): the ranks of a sharded job hold different rows, so their solves take different times and the node runs at the
                             slowest shard's pace -- this measures each shard's pace on the 1-GPU box (tools/scaling_prediction.py runs all of them)
    --one-rank-comm          the library's OWN communicator with world size 1 inside the timed loop: snapshot, speculative solve, ncclAllReduce of the
                             count vector, the host wait for the verdict, ncclAllGather of the packed results on the communication stream -- the exact
                             call sequence of a rank of an N > 1 job (RCCL's launch path, no wire), so its cost over the plain loop is measured
For N > 1 the line also carries per_rank_ms (min / median / max of the ranks' own loop times), gather_ms (event-timed on the communication
stream) and solve_ms_without_gather, so that an efficiency below 1 can be attributed to skew, to the collective, or to RCCL's kernels
taking CUs from the next solve.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
FP32_PEAK_TFLOPS = 157.3   # vector fp32 spec


def stage_bytes(nq, N, fused_schur, fused_step):
    """Algorithmic HBM bytes per trajectory and launch of each kernel family: compulsory reads + writes of the
    buffers that cross the kernel boundary, each buffer counted once per launch (re-reads by the lanes of one launch are cache hits)."""
    nx, nu = 2 * nq, nq
    traj = (nx + nu) * N - nu
    vec = (N + 2) * nx
    nD, nQq, nQd, nq_, nR, nr, nc = (N - 1) * 3 * nq * nq, N * nq * nq, N * nq, N * nx, (N - 1) * nu, (N - 1) * nu, N * nx
    inv = nQq + nQd + nR                      # (Q + rho I)^-1 blocks, R^-1
    lin = nD + inv + nq_ + nr + nc + 1        # what the Schur complement is formed from
    row0 = 2 * nx * nx + nx                   # S / P^-1 main blocks of block row 0 and gamma_0
    nS, nPd = (3 * N - 2) * nx * nx, N * nx * nx
    f = 4
    dz_in = vec + nD + inv + nq_ + nr
    merit_in = 2 * traj + 6 * N + nx + 6 + 1
    out = {
        "kkt": f * (traj + 6 * N + 6 + nx + 1 + nD + nQq + nQd + nq_ + nR + nr + nc + inv + (row0 if fused_schur else 0)),
        "schur": 0 if fused_schur else f * (lin + nq * nq + nq + nS + nPd + vec),
        "pcg": f * ((lin + row0 + vec + 2 + vec + 2) if fused_schur else (nS + nPd + 2 * vec + 2 + vec + 2)),
        "dz": 0 if fused_step else f * (dz_in + traj + nq_ + nr),
        "merit": f * ((dz_in + merit_in + traj + nq_ + nr + traj + 8 + 8) if fused_step else (merit_in + 8)),
        "line_search": 0 if fused_step else f * (3 * traj + 8 + 6),
    }
    out["merit1"] = f * (traj + 6 * N + nx + 6 + 1 + 1)  # the merit of the initial iterate (first launch of a solve)
    return out


def pcg_flops(nq, N, pcg_iters_all, fused_schur):
    """Algorithmic flops of ONE launch of the PCG kernel family over the whole batch (mul + add = 2), counted from the iteration
    counts the device reports -- not an estimate of the reference's dense algebra (SURVEY.md 8(d)'s F_schur etc. are for that):
      per PCG iteration and trajectory: two block-tridiagonal products 2 x 2 (3 nx)(N nx), three axpys 3 x 2 N nx, two dots 2 x 2 N nx
      (pcg.cuh:96-141); before the loop: r = gamma - S x, z = P^-1 r, one dot.
      fused kernel only, per knot: phi = A Q^-1 (q block dense nq^2, qd block diagonal), theta = Q^-1 + phi A^T + (B R^-1) B^T,
      gamma (4 products), the Gauss-Jordan inverse of theta (2 nx^3) and the stair fold (2 products of nx^3)  (schur_linsys.cuh:14-260).
    Returns flops per launch averaged over the launches of the solve."""
    nx, nu = 2 * nq, nq
    rows = N * nx
    mv = 2 * 3 * nx * rows
    per_iter = 2 * mv + 3 * 2 * rows + 2 * 2 * rows
    pre = 2 * mv + 2 * rows
    schur = 0
    if fused_schur:
        phi = 2 * nx * nq * nq + nx * nq
        theta = 2 * nx * nx * nx + 2 * nx * nu * nx + nx * nu
        gam = 2 * nx * (nq + 1 + nx + nu)
        schur = N * (phi + theta + gam + 2 * nx ** 3 + 2 * 2 * nx ** 3)
    it = np.asarray(pcg_iters_all, np.float64)            # [launches][B]
    return float((it.sum(axis=1) * per_iter + it.shape[1] * (pre + schur)).mean())


def source_hash():
    """hash of the kernel sources in THIS tree (tools/source_hash.py); profiles/pmc_summary.json carries the one its counters were measured on, and the
    loaded libgato_hip.so the one it was built from (library_build()): all three go onto the line"""
    from tools.source_hash import source_hash as _h
    return _h()


def library_build():
    """the source hash baked into the libgato_hip.so this process has LOADED (-DGATO_SRC_HASH in gato_amd/csrc/Makefile): a stale binary says so"""
    import ctypes
    from gato_amd import _lib
    L = _lib.load()
    L.gato_source_hash.restype = ctypes.c_char_p
    return L.gato_source_hash().decode()


def usable_cores():
    """Host threads this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_baseline(plant, N, params, dt, sample_b, make_problem, library=None):
    """The CPU oracle (a C port of the reference's algorithm, oracle/gato_oracle.c) on this box's host cores, bounded sample:
    trajectories are independent, so `cores` single-threaded oracle solvers each take a contiguous slice of the sample (no barriers).
    library: another build of the same source (oracle.build_native) instead of the committed one."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleSolver as _OS

    def OracleSolver(*a_, **k_):
        return _OS(*a_, library=library, **k_)
    cores = min(usable_cores(), 128, sample_b)
    per = sample_b // cores
    sample_b = per * cores
    pr = make_problem(sample_b)
    solvers = [OracleSolver(plant, N, per, dt=dt, threads=1, **params) for _ in range(cores)]
    if "rho" in pr:
        for i, sv in enumerate(solvers):
            sv.set_rho_penalty_batch(pr["rho"][i * per:(i + 1) * per], True)

    def run(i):
        sl = slice(i * per, (i + 1) * per)
        solvers[i].reset_dual()
        solvers[i].reset_rho()
        return solvers[i].solve(pr["xu"][sl], dt, pr["x_s"][sl], pr["ref"][sl])["iters_done"]

    import threading
    gate = threading.Barrier(cores)
    warm = [OracleSolver(plant, N, 1, dt=dt, threads=1, **dict(params, max_sqp_iters=1)) for _ in range(cores)]

    def warm_up(i):   # every pool thread enters the library once before the clock starts (thread start-up, OpenMP per-thread init, page faults)
        gate.wait()
        warm[i].solve(pr["xu"][:1], dt, pr["x_s"][:1], pr["ref"][:1])

    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(warm_up, range(cores)))
        t = float("inf")
        for _ in range(3):                            # best of 3 passes: the first pass on a cold host runs several times slower
            t0 = time.perf_counter()
            iters = list(ex.map(run, range(cores)))   # ctypes releases the GIL inside orc_solve
            t = min(t, time.perf_counter() - t0)
    rate = per * sum(iters) / t
    solvers[0].reset_dual()
    solvers[0].reset_rho()
    t0 = time.perf_counter()
    o1 = solvers[0].solve(pr["xu"][:per], dt, pr["x_s"][:per], pr["ref"][:per])
    t1 = time.perf_counter() - t0
    single = per * o1["iters_done"] / t1
    return {"value": rate, "unit": "trajectory-SQP-iterations/s", "cores": cores, "kind": "port",
            "sample": "first %d trajectories of the same workload, %d SQP iterations each, %d single-threaded oracle solvers side by side "
                      "(best of 3 passes, %.2f s); one thread alone: %.0f traj-SQP-iter/s" % (sample_b, iters[0], cores, t, single),
            "single_core_value": single}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--plant", default="indy7")
    ap.add_argument("--knots", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories per GPU")
    ap.add_argument("--sqp-iters", type=int, default=10)
    ap.add_argument("--workload", default="fig8", choices=["fig8", "hparam"],
                    help="fig8: figure-8 tracking windows (the headline, C2 / C4); hparam: BASELINE config C5, rank g solves shard g of the sweep")
    ap.add_argument("--torch-gather", action="store_true", help="N > 1: gather through torch.distributed instead of the library's own communicator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="N > 1 ranks sharing cuda:0 over gloo: exercises the multi-rank code path on a 1-GPU box; the value is not a scaling number")
    ap.add_argument("--as-rank", type=int, default=None, help="with --of G: solve the rows rank R of a G-rank job would hold (one process, one device)")
    ap.add_argument("--of", type=int, default=None, dest="of_ranks")
    ap.add_argument("--one-rank-comm", action="store_true",
                    help="world size 1 through the library's own communicator: the sharded call sequence (deferred count, host wait, ncclAllGather) in the timed loop")
    a = ap.parse_args()
    if (a.as_rank is None) != (a.of_ranks is None) or (a.as_rank is not None and not 0 <= a.as_rank < a.of_ranks):
        raise SystemExit("--as-rank R --of G go together, 0 <= R < G")
    if (a.as_rank is not None or a.one_rank_comm) and a.gpus != 1:
        raise SystemExit("--as-rank / --one-rank-comm are single-process runs on one device (--gpus 1)")

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU fallback)")
    rehearse = bool(a.rehearse_one_device) and world > 1
    if not rehearse and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d wants cuda:%d but %d device(s) are visible: one process per GPU (or --rehearse-one-device to run the multi-rank "
                         "code path on a single GPU)" % (rank, local_rank, torch.cuda.device_count()))
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # bookkeeping collectives (flags, times) run on `cdev`: the device under RCCL, the host under gloo (the rehearsal)
    cdev = torch.device("cpu") if rehearse else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem, hparam_problem
    plant, N, B = a.plant, a.knots, a.batch
    shard_idx = rank if a.as_rank is None else a.as_rank      # whose rows this process solves
    # the CPU baseline is timed on BOTH builds of the oracle -- the committed x86-64-v3 library and one compiled for THIS box's host (-O3
    # -march=native) -- and the faster one is reported, by name (round 4: the native build was the slower one on the driver's box)
    oracle_native = None
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        try:
            import tempfile
            from oracle import oracle as _orc
            oracle_native = _orc.build_native(tempfile.mkdtemp(prefix="gato_oracle_"))
        except Exception:   # noqa: BLE001
            oracle_native = None
    if a.workload == "hparam":
        # BASELINE config C5: rank g = shard g of the sweep (cost tuple g, per-trajectory rho, dt 0.05, mu 1, pcg_tol 1e-3)
        def make_problem(n, shard=shard_idx):
            return hparam_problem(plant, N, n, shard=shard)
        pr = make_problem(B)
        params, dt = dict(pr["params"], max_sqp_iters=a.sqp_iters), pr["dt"]
        what = ("hyper-parameter sweep (gato_hparam_batch.ipynb): one random goal per trajectory, shard g on rank g = cost tuple g of the grid, "
                "per-trajectory rho 1e-8..1e1, dt 0.05, mu 1, pcg_tol 1e-3")
    else:
        def make_problem(n, offset=shard_idx * B):
            return fig8_problem(plant, N, n, batch_offset=offset)      # rank r owns rows [r*B, (r+1)*B) of the global batch
        pr = make_problem(B)
        params, dt = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=a.sqp_iters), 0.01
        what = "figure-8 end-effector tracking, DEFAULT_SOLVER_PARAMS (max_pcg 200, pcg_tol 1e-4, rho 0.01)"
    solver = NativeSolver(plant, N, B, dt=dt, **params)
    if "rho" in pr:
        solver.set_rho_penalty_batch(pr["rho"], True)      # also the value reset_rho() goes back to
    from gato_amd.sharding import PackedResults, check_sharded_params, connect
    native = False
    collective = "none"
    if world > 1:
        # the library's own communicator (gato_comm_init): the solved count of the exit rule is shared per SQP iteration inside the solve
        # and the results travel by ncclAllGather on it.  Checked before use: every rank gathers a known pattern; if ANY rank fails the
        # whole job falls back to torch.distributed for the results (and to per-shard counting, exact here: fig-8 / sweep rows do not converge)
        ok_t = torch.ones(1, device=cdev)
        try:
            connect(solver)   # agrees on RCCL's availability across the ranks BEFORE any rank enters ncclCommInitRank: fails on all ranks or none
            probe = torch.full((4,), float(rank + 1), device=dev)
            got = torch.zeros(4 * world, device=dev)
            solver.gather_results(probe.data_ptr(), got.data_ptr(), 4, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want = torch.arange(1, world + 1, device=dev, dtype=torch.float32).repeat_interleave(4)
            if not torch.equal(got, want):
                ok_t.zero_()
        except Exception as e:   # noqa: BLE001
            print("rank %d: native communicator unavailable (%s)" % (rank, e), file=sys.stderr)
            ok_t.zero_()
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        coupled = bool(ok_t.item() > 0)
        native = coupled and not a.torch_gather
        if coupled:
            check_sharded_params(params["solve_ratio"], world, coupled=True)
        else:
            # explicit fallback: no rank has a usable communicator.  The shards count their own rows against the WHOLE batch's threshold (the
            # other shards' counts taken as zero): exact as long as the whole batch's rule never fires -- verified after the run (`checks` below)
            try:
                solver.comm_destroy()
            except Exception:   # noqa: BLE001
                pass
            solver.debug_set_remote_solved(np.zeros(1, np.uint32), world * B)
        collective = "ncclAllGather on the library's communicator" if native else "torch.distributed.all_gather_into_tensor"
    one_rank = bool(a.one_rank_comm)
    if one_rank:
        # the library's own communicator, world size 1: every call of a rank of an N > 1 job below (no torch.distributed in this process)
        solver.comm_init(solver.comm_unique_id(), 1, 0)
        probe = torch.full((4,), 1.0, device=dev)
        got = torch.zeros(4, device=dev)
        solver.gather_results(probe.data_ptr(), got.data_ptr(), 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if not torch.equal(got, probe):
            raise SystemExit("--one-rank-comm: the one-rank ncclAllGather did not return the probe")
        native, coupled = True, True
        collective = "ncclAllGather on the library's communicator (ONE rank: RCCL's launch path, no wire)"
    sharded_loop = world > 1 or one_rank
    xu0 = torch.from_numpy(pr["xu"]).to(dev)
    x_s = torch.from_numpy(pr["x_s"]).to(dev)
    ref = torch.from_numpy(pr["ref"]).to(dev)
    # two packed result buffers [B*TRAJ iterates | B merits], solved in place and gathered by ONE collective each: the gather of solve n
    # runs on a communication stream while solve n+1 iterates in the other buffer (collectives overlapped with compute on separate HIP
    # streams); a buffer is reused only after the gather that read it has completed (per-buffer events)
    pks = [PackedResults(B, solver.traj, world, dev, own_image=one_rank) for _ in range(2)]
    main = torch.cuda.current_stream()
    stream = main.cuda_stream
    comm = torch.cuda.Stream() if sharded_loop else None
    done = [torch.cuda.Event(), torch.cuda.Event()]
    used = [False, False]
    count = [0]

    gather_ev = []   # (start, end) events on the communication stream, one pair per gather
    ready = [torch.cuda.Event(), torch.cuda.Event()]   # buffer j holds the finished results of its solve (recorded on the main stream)
    owed = []        # the buffer whose gather has not been enqueued yet (at most one)

    def enqueue_gather():
        """the ONE data-path collective of the solve that last finished enqueueing, on the communication stream behind that solve's `ready` event"""
        if not owed:
            return
        j = owed.pop()
        comm.wait_event(ready[j])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(comm):
            e0.record(comm)
            pks[j].all_gather(solver=solver if native else None, stream=comm.cuda_stream)   # iterates + costs: (B x TRAJ + B) fp32 per rank over xGMI
            e1.record(comm)
        gather_ev.append((e0, e1))
        done[j].record(comm)
        used[j] = True

    def step(gather=True):
        """one pass of the hot path.  Sharded, the library leaves the verdict of the (speculative) solve to its next entry point: between the solve's
        launches and that entry point the host has nothing to wait for, and that is where the PREVIOUS solve's gather is enqueued -- with the device
        busy, not idle behind a host wait.  The gather of solve n therefore runs on the communication stream beside solve n + 1."""
        j = count[0] & 1
        count[0] += 1
        pk = pks[j]
        if used[j]:
            main.wait_event(done[j])            # the gather that last read this buffer
        solver.reset_async(True, True, stream)  # reset_dual() + reset_rho(), stream-ordered
        pk.xu.copy_(xu0)
        solver.solve_device(pk.xu.data_ptr(), dt, x_s.data_ptr(), ref.data_ptr(), stream)
        if comm is not None:
            enqueue_gather()                    # of the previous step (its results are final: its verdict was taken by its merit copy)
        solver.copy_final_merit_device(pk.merit.data_ptr(), stream)   # takes this solve's verdict first (sharded: the one host wait of the step)
        if comm is not None and gather:
            ready[j].record(main)
            owed.append(j)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # which rows this rank solves, said by the rank itself: its first reference point and first start state go onto the line (multi_gpu.shards)
        mine = torch.tensor([float(rank), float(pr["ref"][0, 0]), float(pr["ref"][0, 1]), float(pr["ref"][0, 2]), float(pr["x_s"][0, 0]),
                             float(params["q_cost"]), float(params["qd_cost"]), float(params["u_cost"]), float(params["N_cost"])], dtype=torch.float64, device=cdev)
        shard_rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(shard_rows, mine)
        shard_rows = torch.stack(shard_rows).cpu().numpy()
    for _ in range(a.warmup):
        step()
    if comm is not None:
        enqueue_gather()
    sync()
    stage_acc = {}
    gather_ev.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if comm is not None:
        enqueue_gather()                      # the last solve's gather belongs to the timed region: K solves, K gathers
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0          # this rank's own loop, before it waits for the others
    sync()
    t = time.perf_counter() - t0
    multi = {}
    if sharded_loop:
        def over_ranks(vals, op=None):
            """[world][len(vals)] float64: every rank's values (one process: its own)"""
            v = torch.tensor([float(x) for x in vals], dtype=torch.float64, device=cdev)
            if world == 1:
                return v.cpu().numpy()[None, :]
            rows = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(rows, v)
            return torch.stack(rows).cpu().numpy()
        t = float(over_ranks([t])[:, 0].max())
        g_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in gather_ev])) if gather_ev else 0.0
        own = over_ranks([1e3 * t_own / a.steps, g_ms])
        # the same loop with the gather switched off (untimed for the headline): what the collective and its kernels cost the solves
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(max(2, a.steps // 4)):
            step(gather=False)
        solver.synchronize()   # (takes the last verdict)
        e1.record(main)
        sync()
        ng = float(over_ranks([e0.elapsed_time(e1) / max(2, a.steps // 4)])[:, 0].max())
        # how the solves of this run actually learned the count, from the LIBRARY (not from the environment): speculative solves, the replays among
        # them, and the solves that counted per iteration (the mode, or the back-off after a replay) -- summed over the ranks
        sh, cs = solver.shard_stats(), solver.solved_count_state()
        cnt = over_ranks([sh["deferred_solves"], sh["replays"], cs["per_iteration_solves"], 1.0 if cs["mode"] == "deferred" else 0.0])
        modes = sorted({"deferred" if m else "per_iteration" for m in cnt[:, 3]})
        multi = {"collective": collective,
                 "solved_count": ("/".join(modes) + " (gato_get_solved_count_state)") if coupled else "per shard (no communicator)",
                 "solved_count_detail": ("over all ranks: %d speculative solves (ONE ncclAllReduce of the per-iteration count vector each), %d of them replayed exactly, "
                                         "%d solves with one 4-byte ncclAllReduce per SQP iteration (the mode, or the back-off after a replay)"
                                         % (cnt[:, 0].sum(), cnt[:, 1].sum(), cnt[:, 2].sum())) if coupled
                                        else "shards count alone: exact while the whole batch's exit rule never fires (checked: solution_checks)",
                 "solves_by_count_form": {"speculative": int(cnt[:, 0].sum()), "replayed": int(cnt[:, 1].sum()), "per_iteration": int(cnt[:, 2].sum())},
                 "per_rank_ms": {"min": float(own[:, 0].min()), "median": float(np.median(own[:, 0])), "max": float(own[:, 0].max())},
                 "gather_ms": {"mean_over_ranks": float(own[:, 1].mean()), "max_over_ranks": float(own[:, 1].max())},
                 "solve_ms_without_gather": ng}
        if world > 1:
            multi["shards"] = [{"rank": int(r[0]), "first_ref_xyz": [float(r[1]), float(r[2]), float(r[3])], "first_q0": float(r[4]),
                                "cost_tuple": {"q_cost": float(r[5]), "qd_cost": float(r[6]), "u_cost": float(r[7]), "N_cost": float(r[8])}} for r in shard_rows]
        if rehearse:
            multi["rehearsal"] = ("%d ranks sharing ONE device over gloo: a run of the multi-rank code path, not a scaling measurement "
                                  "(the solves of the ranks time-share the GPU)" % world)
        if one_rank:
            multi["one_rank_comm"] = ("ONE rank through the library's own communicator: snapshot + speculative solve + ncclAllReduce of the count vector + the host "
                                      "wait for the verdict + ncclAllGather on the communication stream in every timed step (a rank's call sequence in an N > 1 job)")

    st = solver.stats()
    iters = st["iters_done"]
    # kernel durations measured live: one extra (untimed) step with hipEvents on the solver's stream around each kernel family
    solver.set_profiling(True)
    step()
    if comm is not None:
        enqueue_gather()
    sync()
    stage_acc = solver.stage_times_us()
    solver.set_profiling(False)
    # every trajectory finite and none worse than it started (a sweep row whose rho makes every line search fail keeps its merit: equal)
    checks = {"finite": bool(np.all(np.isfinite(st["final_merit"]))), "no_row_worse_than_it_started": bool(np.all(st["final_merit"] <= st["initial_merit"])),
              "some_row_improved": bool(np.any(st["final_merit"] < st["initial_merit"]))}
    if world > 1 and not coupled:
        # shards that count alone (the other shards' counts taken as zero against the WHOLE batch's threshold) never fire the rule; with solve_ratio = 1
        # the true rule fires only once EVERY row of EVERY shard is converged, so a shard that still holds an unconverged row at the end proves the
        # two agree (convergence flags are sticky: bsqp.cuh:153-156)
        checks["uncoupled_shard_kept_an_unconverged_row"] = bool(params["solve_ratio"] >= 1.0 and not np.all(st["kkt_converged"]))
    ok = all(checks.values())
    # a parity bit on the line itself (untimed tail; the oracle is the CHECKER here, never the thing measured): 16 rows spread over this rank's
    # batch, solved by the CPU oracle from the same reset state -- the first SQP iteration's decisions (line-search step, PCG count +-1, initial
    # merit) of the timed solves must be the oracle's.  Later iterations of a free-running fp32 solve are not comparable row by row.
    parity = None
    if rank == 0 and iters > 0 and st["ls_num_iters"] > 0:
        try:
            from oracle.oracle import OracleSolver
            idx = np.unique(np.linspace(0, B - 1, 16).astype(int))
            orc = OracleSolver(plant, N, len(idx), dt=dt, threads=min(usable_cores(), len(idx)), **dict(params, max_sqp_iters=1))
            if "rho" in pr:
                orc.set_rho_penalty_batch(pr["rho"][idx], True)
            ro = orc.solve(pr["xu"][idx], dt, pr["x_s"][idx], pr["ref"][idx])
            # a row passes when it takes the oracle's step, or a step the oracle's OWN merits cannot tell from it (within 2e-2: one decision to fp32)
            sg, so = st["ls_step_size"][0][idx], ro["ls_step_size"][0]
            passed = 0
            for j in range(len(idx)):
                cand = {**{float(2.0 ** -i): float(ro["ls_merits"][0, j, i]) for i in range(8)}, -1.0: float(ro["ls_merit_before"][0, j])}
                passed += int(sg[j] == so[j] or abs(cand[float(sg[j])] - cand[float(so[j])]) <= 2e-2 * max(1.0, abs(cand[float(so[j])])))
            pcg_n = int((np.abs(ro["pcg_iters"][0].astype(int) - st["pcg_iters_all"][0][idx]) <= 1).sum())
            im = float(np.abs(ro["initial_merit"] - st["initial_merit"][idx]).max() / max(1e-30, np.abs(ro["initial_merit"]).max()))
            # figure-8 rows are within fp32's reach: every sampled row must pass.  The hyper-parameter sweep is not (the fp32 oracle itself is
            # 1.5e-2 from its float64 build after one iteration and takes the float64 step on ~3 rows in 4, tests/test_full_size_oracle_gpu.py):
            # there the sample is reported, and only a majority is required.
            need = len(idx) if a.workload == "fig8" else len(idx) // 2
            parity = {"rows": int(len(idx)), "checker": "oracle/gato_oracle.c (fp32), first SQP iteration from the reset state",
                      "rows_on_the_oracles_step_or_a_tie": passed, "rows_pcg_iters_within_1": pcg_n, "rows_required": need, "initial_merit_rel_err": im}
            checks["parity_sample"] = bool(passed >= need and pcg_n >= need and im < 1e-5)
            ok = ok and checks["parity_sample"]
        except Exception as e:   # noqa: BLE001
            parity = {"error": "%s: %s" % (type(e).__name__, e)}
            checks["parity_sample"] = False
            ok = False
    if world > 1:
        # the line carries the AND over ALL shards (a non-finite or worsened row on rank 5 must reach it), and says which ranks failed what
        names = sorted(k for k in checks if k != "parity_sample")            # the parity sample is rank 0's alone
        fl = torch.tensor([1.0 if checks[k] else 0.0 for k in names], dtype=torch.float64, device=cdev)
        rows = [torch.zeros_like(fl) for _ in range(world)]
        dist.all_gather(rows, fl)
        rows = torch.stack(rows).cpu().numpy()
        failed = {k: [int(r) for r in np.nonzero(rows[:, i] == 0)[0]] for i, k in enumerate(names) if (rows[:, i] == 0).any()}
        for i, k in enumerate(names):
            checks[k] = bool(rows[:, i].all())
        if failed:
            checks["failed_on_ranks"] = failed
        ok = all(v for k, v in checks.items() if k != "failed_on_ranks")
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = world * B * iters * a.steps / t
    fused_schur = stage_acc.get("schur", 0.0) == 0.0
    fused_step = stage_acc.get("dz", 0.0) == 0.0
    launches = {"kkt": iters, "schur": 0 if fused_schur else iters, "pcg": iters, "dz": 0 if fused_step else iters, "merit": iters + 1,
                "line_search": 0 if fused_step else iters}
    per_launch_us = {k: (stage_acc[k] / launches[k] if launches[k] else 0.0) for k in launches}
    dom = max(per_launch_us, key=lambda k: stage_acc[k])
    sb = stage_bytes(NativeSolver_nq(plant), N, fused_schur, fused_step)
    if dom == "merit":   # iters step launches + the initial merit share the stage clock
        dom_bytes = (sb["merit"] * iters + sb["merit1"]) / (iters + 1) * B
    else:
        dom_bytes = sb[dom] * B
    achieved = dom_bytes / (per_launch_us[dom] * 1e-6) / 1e9
    iter_bytes = sum(sb[k] for k in ("kkt", "schur", "pcg", "dz", "merit", "line_search"))  # per trajectory and SQP iteration
    # both roofs of the dominant kernel (the PCG launch): HBM from the algorithmic bytes, fp32 VALU from the flops the device's own
    # iteration counts imply; `bound` is the larger fraction.  The launch is latency-shaped: both are small.
    hbm_frac = achieved / HBM_PEAK_GBS
    # the binary that ran vs the sources of this tree: the library carries the hash of what it was built from (gato_source_hash); a stale .so fails the line
    lib_build = library_build()
    checks["library_built_from_this_tree"] = bool(lib_build == source_hash())
    ok = ok and checks["library_built_from_this_tree"]
    valu = None
    if dom == "pcg":
        fl = pcg_flops(NativeSolver_nq(plant), N, st["pcg_iters_all"], fused_schur)
        tf = fl / (per_launch_us[dom] * 1e-6) / 1e12
        valu = {"achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "algorithmic_flops_per_launch": fl}
    # counter-measured traffic and issue-side counters of the profiled build (tools/profile_round.sh + tools/summarize_pmc.py)
    traffic, pmc = None, {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path):
        try:
            js = json.load(open(pmc_path))
            pat = {"pcg": "pcg", "kkt": "kkt_kernel", "merit": "step_kernel"}.get(dom, dom)
            section = {("indy7", 32, 1024, "fig8"): "c2", ("iiwa14", 128, 256, "fig8"): "c3", ("iiwa14", 64, 512, "hparam"): "c5"}.get((plant, N, B, a.workload))
            rows = [(k, v) for k, v in js.get(section, {}).items() if pat in k] if section else []   # counters only of the configuration that ran
            if rows:
                k, v = max(rows, key=lambda kv: kv[1].get("pct_of_gpu_time", 0.0))
                traffic = v.get("hbm_bytes")
                pmc = {"section": section, "kernel": k, "profiled_build": js.get("build"), "this_build": source_hash(), "library_build": lib_build,
                       "build_matches": js.get("build") == source_hash() == lib_build,
                       "rocprof_avg_us": v.get("avg_us"), "valu_issue_frac": v.get("valu_issue_frac"), "lds_bank_conflict_frac": v.get("lds_bank_conflict_frac"),
                       "mfma_busy_cycles": v.get("mfma_busy_cycles")}
        except Exception:
            traffic, pmc = None, {}
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac, "algorithmic_bytes_per_launch": dom_bytes}
    top = valu if (valu and valu["frac"] >= hbm_frac) else hbm
    line = {
        "metric": ("REHEARSAL on one device (not a measurement): " if rehearse else "")
                  + ("SQP iterations/sec (whole node), indy7 N=32 batch=1024, 1/2/4/8 MI355X" if (plant, N, B, a.workload) == ("indy7", 32, 1024, "fig8")
                     else "SQP iterations/sec (whole node), %s N=%d batch=%d %s (not the headline configuration)" % (plant, N, B, a.workload))
                  + (" -- the rows of rank %d of %d, solved alone on one device (scaling prediction, not the headline)" % (a.as_rank, a.of_ranks) if a.as_rank is not None else "")
                  + (" -- through the library's one-rank communicator" if one_rank else ""),
        "value": value, "unit": "trajectory-SQP-iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * t / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s N=%d batch=%d per GPU (global %d), %s, %d SQP iterations per solve, reset_dual+reset_rho per solve"
                               % (plant, N, B, world * B, what, iters),
                   "plant": plant, "knot_points": N, "batch_per_gpu": B, "global_batch": world * B, "sqp_iters_per_solve": int(iters),
                   "rows_of_rank": {"rank": int(shard_idx), "of": int(a.of_ranks if a.as_rank is not None else world)},
                   "sum_over_launches_of_max_pcg_iters": int(st["pcg_iters_all"].max(axis=1).sum()),
                   "mean_pcg_iters": float(st["pcg_iters_all"].mean()), "parallelism": ("one GPU: the whole batch in one solver handle" if world == 1 else
                                   "batch-sharded x%d (one process per GPU), the solved count of the exit rule reduced ONCE per solve behind a speculative solve "
                                   "(multi_gpu.solved_count), one packed all_gather of iterates + merits per solve, overlapped with the next solve" % world)},
        "library": {"version": lib_version(), "built_from": lib_build, "tree": source_hash()},
        "roofline": {"bound": "valu" if top is valu else "hbm", "kernel": dom, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
                     "frac": top["frac"], "traffic": traffic, "hbm": hbm, "valu": valu, "pmc": pmc, "avg_launch_us": per_launch_us[dom],
                     "stage_us_per_solve": {k: round(v, 1) for k, v in stage_acc.items()},
                     "kernels_per_sqp_iteration": sum(1 for k in launches if launches[k] and k != "merit") + 1,
                     "whole_iteration": {"algorithmic_bytes_per_traj_iter": iter_bytes,
                                         "hbm_frac": iter_bytes * value / world / 1e9 / HBM_PEAK_GBS}},
        "solution_ok": ok,
        "solution_checks": checks,
        "parity_sample": parity,
    }
    if multi:
        line["multi_gpu"] = multi
    if not a.no_cpu_baseline and world == 1:
        builds = {"x86-64-v3 (the committed oracle/libgato_oracle.so)": cpu_baseline(plant, N, params, dt, a.cpu_sample, make_problem)}
        if oracle_native:
            try:
                builds["-O3 -march=native, built on this host"] = cpu_baseline(plant, N, params, dt, a.cpu_sample, make_problem, library=oracle_native)
            except Exception:   # noqa: BLE001
                pass
        best = max(builds, key=lambda k: builds[k]["value"])
        line["cpu_baseline"] = builds[best]
        line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        line["cpu_baseline"]["build"] = best
        line["cpu_baseline"]["builds_timed"] = {k: v["value"] for k, v in builds.items()}
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def lib_version():
    from gato_amd import _lib
    return _lib.load().gato_version().decode()


def NativeSolver_nq(plant):
    return {"indy7": 6, "iiwa14": 7}[plant]


if __name__ == "__main__":
    main()
