"""Multi-GPU layer: a batch of independent trajectories sharded over the ranks of one node.

Trajectories never read each other (only the host-side early-exit count couples them, bsqp.cuh:165), so rank r solves rows
[r*B_local, (r+1)*B_local) of the global batch with its own solver handle (its own lambda / rho / f_ext slices) and ONE collective
per solve gathers the iterates and merits: `all_gather` over RCCL/xGMI on GPUs (backend "nccl"), gloo on CPU for the tests.
With solve_ratio = 1 (every shipped configuration) the early exit only fires when ALL trajectories converged; shards then stop
independently, which changes no iterate (a converged shard's extra iterations are what the reference would also have executed
while other trajectories were unconverged).  solve_ratio < 1 across ranks would need a 4-byte SUM all-reduce per iteration and is
not supported sharded (ValueError).
"""
import numpy as np


def shard_bounds(global_batch, world_size, rank):
    if global_batch % world_size:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def gather_results(local, group=None, device=None):
    """all_gather of per-trajectory results. `local`: dict name -> array/tensor with leading dim B_local. Returns dict of
    numpy arrays with leading dim B_global, identical on every rank."""
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


def best_trajectory(final_merit_local, rank, group=None, device=None):
    """MINLOC over the whole sharded batch: (merit, global index) of the best trajectory (MPC selection, mpc_controller.py:294-309)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    fm = np.asarray(final_merit_local, np.float32)
    i = int(np.argmin(fm))
    t = torch.tensor([float(fm[i]), float(rank * fm.size + i)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t, group=group)
    allv = torch.stack(bufs).cpu().numpy()
    j = int(np.argmin(allv[:, 0]))
    return float(allv[j, 0]), int(allv[j, 1])
