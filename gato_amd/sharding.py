"""Multi-GPU layer: a batch of independent trajectories sharded over the ranks of one node.

Trajectories never read each other (only the early-exit count couples them, bsqp.cuh:165), so rank r solves rows
[r*B_local, (r+1)*B_local) of the global batch with its own solver handle (its own lambda / rho / f_ext slices) and ONE collective
per solve gathers iterates and merits: rank-local results live in one packed device buffer [B_local*TRAJ | B_local] that the solver
writes in place, and a single `all_gather_into_tensor` (RCCL over xGMI with backend "nccl"; gloo on CPU for the tests) assembles the
[world][B_local*TRAJ + B_local] image on every rank -- no host round trip, no per-field collectives.

With solve_ratio = 1 (every shipped configuration) the early exit only fires when ALL trajectories of a shard converged; shards then
stop independently, which changes no iterate of an unconverged trajectory (a converged shard's extra iterations are what the reference
would also execute while other trajectories are unconverged).  solve_ratio < 1 couples the shards through the solved count -- it
would need a 4-byte SUM all-reduce per SQP iteration -- and is refused here (`check_sharded_params`), not silently approximated.
"""
import numpy as np


def shard_bounds(global_batch, world_size, rank):
    if global_batch % world_size:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def check_sharded_params(solve_ratio, world_size):
    """Each rank applies the early-exit threshold B*solve_ratio to its LOCAL batch; that equals the unsharded rule only for
    solve_ratio >= 1 (exit when everything converged)."""
    if world_size > 1 and float(solve_ratio) < 1.0:
        raise ValueError("solve_ratio=%g < 1 is not supported on a sharded batch (world size %d): the early exit counts solved trajectories "
                         "over the whole batch (bsqp.cuh:165)" % (solve_ratio, world_size))


class PackedResults:
    """Rank-local result buffer [B*TRAJ iterates | B merits] + its gathered image, both resident on `device`."""

    def __init__(self, B_local, traj, world_size, device="cpu"):
        import torch
        self.B, self.traj, self.world = int(B_local), int(traj), int(world_size)
        self.n = self.B * self.traj + self.B
        self.local = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.gathered = torch.zeros(self.world * self.n, dtype=torch.float32, device=device) if self.world > 1 else self.local

    @property
    def xu(self):       # [B, TRAJ] view the solver iterates in place
        return self.local[: self.B * self.traj].view(self.B, self.traj)

    @property
    def merit(self):    # [B] view the final merits are copied into (device to device)
        return self.local[self.B * self.traj:]

    def all_gather(self, group=None):
        """the ONE collective of a solve; stream-ordered on the current stream for nccl"""
        if self.world > 1:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.gathered, self.local, group=group)
        return self.gathered

    def global_xu(self):     # [world*B, TRAJ]
        g = self.gathered.view(self.world, self.n)
        return g[:, : self.B * self.traj].reshape(self.world * self.B, self.traj)

    def global_merit(self):  # [world*B]
        g = self.gathered.view(self.world, self.n)
        return g[:, self.B * self.traj:].reshape(self.world * self.B)

    def best(self):
        """(merit, global index) of the best trajectory of the whole sharded batch -- the MPC selection of
        mpc_controller.py:294-309 needs no second collective: every rank holds all merits after `all_gather`."""
        m = self.global_merit()
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
