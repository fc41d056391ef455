"""Multi-GPU layer: one batch of independent trajectories sharded over the ranks of a node.

Rank r solves rows [r*B_local, (r+1)*B_local) of the global batch with its own solver handle (its own lambda / rho / f_ext slices).  Two things
cross ranks:

* the SOLVED COUNT of the exit rule (bsqp.cuh:165) -- the only coupling between trajectories.  `connect()` gives the rank's solver a native RCCL
  communicator (`gato_comm_init`, include/gato_abi.h).  By default the count is DEFERRED (round 4): a solve runs speculatively as if the rule
  never fired, ONE ncclAllReduce of the per-iteration count vector follows it, and only when some iteration's whole-batch count reached the
  threshold is the solve restored from its snapshot and re-run with one 4-byte all-reduce per SQP iteration (`set_solved_count_mode`).  Either
  way every rank takes the exit of the WHOLE batch in the same iteration, for any solve_ratio.  That also covers solve_ratio = 1: a shard
  whose rows have all converged keeps stepping them while another shard has not (converged trajectories are still moved by the line search,
  bsqp.cuh:165-171), exactly as the unsharded solver would -- per-shard counting would stop that shard early and make its iterates depend on
  the world size.  Without a communicator (`check_sharded_params`) only the case where nothing can converge differently is accepted: it is refused.
* the RESULTS: rank-local results live in ONE packed device buffer [B_local*TRAJ iterates | B_local merits] that the solver writes in place, and
  ONE all-gather per solve assembles the [world][B_local*TRAJ + B_local] image on every rank -- `gato_gather_results` (ncclAllGather on the
  solver's communicator) when the solver has one, `torch.distributed.all_gather_into_tensor` otherwise (gloo on CPU for the tests).
"""
import numpy as np


def shard_bounds(global_batch, world_size, rank):
    if global_batch % world_size:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def check_sharded_params(solve_ratio, world_size, coupled=False):
    """A sharded solve is exact only when the ranks share the solved count (`coupled`: connect() below, or the oracle's set_shard in the tests).
    Uncoupled, each rank would apply the exit rule to its local rows: wrong for solve_ratio < 1, and for solve_ratio >= 1 wrong as soon as one
    shard's rows all converge before another's (that shard would stop stepping its converged rows).  Refused, not approximated."""
    if world_size > 1 and not coupled:
        raise ValueError("a batch sharded over %d ranks needs the solved count shared between them (gato_amd.sharding.connect): the exit rule "
                         "counts solved trajectories over the whole batch (bsqp.cuh:165; solve_ratio=%g)" % (world_size, solve_ratio))


def connect(solver, group=None, timeout_s=120.0):
    """Collective over the ranks of `group`: gives `solver` (a gato_amd._lib.NativeSolver on this rank's device) the native RCCL communicator of
    the sharded batch.  The 128-byte id travels through torch.distributed (any backend); everything after that is RCCL inside the library.
    Fails on EVERY rank or on none: RCCL missing (step 1), ncclCommInitRank failing on one rank (step 3) and the count-mode agreement (step 4)
    are each compared across the ranks over torch.distributed before the next step; on a failure every rank drops its communicator and raises."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # 1. agree that EVERY rank can open RCCL before any rank enters a collective of its own: gato_comm_available proves librccl.so loads and
    #    has the entry points, without calling into it (ncclGetUniqueId would start a bootstrap listener on every rank; only rank 0 needs one).
    #    A rank that fails here must not leave the others blocked in the broadcast or inside ncclCommInitRank.
    uid, err = None, None
    try:
        err = solver.comm_available()
        if err is None and rank == 0:
            uid = solver.comm_unique_id()
    except Exception as e:   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    flags = [None] * world
    dist.all_gather_object(flags, err, group=group)
    bad = {r: f for r, f in enumerate(flags) if f is not None}
    if bad:
        raise RuntimeError("RCCL is not available on rank(s) %s -- no rank initialises a communicator" % bad)   # raised on every rank alike
    # 2. rank 0's id to everybody (rank 0 always broadcasts), then ncclCommInitRank ALONE (gato_comm_init_rank: no collective on the new communicator)
    box = [uid if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    err = None
    try:
        solver.comm_init_rank(box[0], world, rank)
    except Exception as e:   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    # 3. the ranks compare their outcomes over torch.distributed BEFORE anyone issues a collective on the new communicator: a rank whose
    #    initialisation failed after the probe passed (a sick device, RCCL refusing its arguments) would otherwise leave the others inside the
    #    first all-reduce for good.  Bounded: a rank that never arrives (it died) turns into an error here, not a hang.
    bad = _agree(err, world, group, timeout_s, "gato_comm_init_rank")
    if bad:
        _drop(solver)
        raise RuntimeError("the communicator could not be initialised on rank(s) %s -- every rank drops it" % bad)
    # 4. the first collective on the new communicator: the ranks agree on the solved-count mode (a disagreement fails on every rank alike)
    try:
        solver.comm_confirm()
    except Exception as e:   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    bad = _agree(err, world, group, timeout_s, "gato_comm_confirm")
    if bad:
        _drop(solver)
        raise RuntimeError("the ranks did not agree on the solved-count mode / rank(s) %s failed -- every rank drops the communicator" % bad)
    return solver


def _drop(solver):
    try:
        solver.comm_destroy()
    except Exception:   # noqa: BLE001
        pass


def _agree(err, world, group, timeout_s, what):
    """{rank: reason} of the ranks that reported a failure (empty: none), the same on every rank; a rank that does not report within timeout_s
    counts as failed on the ranks that waited for it."""
    import datetime
    import torch
    import torch.distributed as dist
    # a fixed-size tensor (not an object collective): works asynchronously on every backend, so the wait can be bounded
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.zeros(257, dtype=torch.uint8)
    if err is not None:
        raw = err.encode("utf-8", "replace")[:256]
        mine[0] = 1
        mine[1:1 + len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    mine = mine.to(dev)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    try:
        work = dist.all_gather(rows, mine, group=group, async_op=True)
        if not work.wait(datetime.timedelta(seconds=timeout_s)):
            raise RuntimeError("timed out")
        if dev == "cuda":
            torch.cuda.current_stream().synchronize()
    except Exception as e:   # noqa: BLE001
        raise RuntimeError("a rank did not report its %s outcome within %.0f s (%s): the job cannot continue" % (what, timeout_s, e))
    bad = {}
    for r, row in enumerate(rows):
        row = row.cpu()
        if int(row[0]):
            bad[r] = bytes(row[1:].tolist()).rstrip(b"\0").decode("utf-8", "replace")
    return bad


class PackedResults:
    """Rank-local result buffer [B*TRAJ iterates | B merits] + its gathered image, both resident on `device`."""

    def __init__(self, B_local, traj, world_size, device="cpu", own_image=False):
        """own_image: a gathered image of its own even for ONE rank (a one-rank communicator then really runs its ncclAllGather: bench.py --one-rank-comm)"""
        import torch
        self.B, self.traj, self.world = int(B_local), int(traj), int(world_size)
        self.n = self.B * self.traj + self.B
        self.local = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.gathered = torch.zeros(self.world * self.n, dtype=torch.float32, device=device) if (self.world > 1 or own_image) else self.local

    @property
    def xu(self):       # [B, TRAJ] view the solver iterates in place
        return self.local[: self.B * self.traj].view(self.B, self.traj)

    @property
    def merit(self):    # [B] view the final merits are copied into (device to device)
        return self.local[self.B * self.traj:]

    def all_gather(self, group=None, solver=None, stream=0):
        """the ONE data-path collective of a solve.  solver (connected): ncclAllGather on its communicator, enqueued on `stream` (a raw
        hipStream_t); otherwise torch.distributed on torch's current stream."""
        if self.world > 1 or (solver is not None and self.gathered is not self.local):
            if solver is not None:
                solver.gather_results(self.local.data_ptr(), self.gathered.data_ptr(), self.n, stream)
            else:
                import torch
                import torch.distributed as dist
                if self.local.is_cuda and dist.get_backend(group) == "gloo":
                    # device buffers, host collective (no RCCL between the ranks: two ranks sharing one GPU, or a node whose RCCL does not load):
                    # staged through pinned host memory, ordered after the solve on torch's current stream
                    if getattr(self, "_h_local", None) is None:
                        self._h_local = torch.empty(self.n, dtype=self.local.dtype, pin_memory=True)
                        self._h_all = torch.empty(self.world * self.n, dtype=self.local.dtype, pin_memory=True)
                    self._h_local.copy_(self.local, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    dist.all_gather_into_tensor(self._h_all, self._h_local, group=group)
                    self.gathered.copy_(self._h_all, non_blocking=True)
                else:
                    dist.all_gather_into_tensor(self.gathered, self.local, group=group)
        return self.gathered

    def global_xu(self):     # [world*B, TRAJ]
        g = self.gathered.view(self.world, self.n)
        return g[:, : self.B * self.traj].reshape(self.world * self.B, self.traj)

    def global_merit(self):  # [world*B]
        g = self.gathered.view(self.world, self.n)
        return g[:, self.B * self.traj:].reshape(self.world * self.B)

    def best(self):
        """(merit, global index) of the trajectory with the lowest FINAL MERIT of the whole sharded batch -- every rank holds all merits after
        `all_gather`, so this needs no second collective.  (This is merit-based selection, e.g. for a hyper-parameter sweep; the MPC loop's
        hypothesis selection is by prediction error, gato_select_best.)  Non-finite merits never win."""
        import torch
        m = self.global_merit()
        m = torch.where(torch.isfinite(m), m, torch.full_like(m, float("inf")))
        i = int(m.argmin().item())
        return float(m[i].item()), i


def gather_results(local, group=None, device=None):
    """Generic all_gather of per-trajectory arrays (statistics, not the per-solve data path -- that is PackedResults).
    `local`: dict name -> array/tensor with leading dim B_local; returns numpy arrays with leading dim B_global on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = {}
    for name, val in local.items():
        t = val if isinstance(val, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(val))
        if device is not None:
            t = t.to(device)
        t = t.contiguous()
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t, group=group)
        out[name] = torch.cat(bufs, dim=0).cpu().numpy()
    return out
