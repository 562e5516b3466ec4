"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

  * extract + match: frames/streams are independent -> `shard_streams`; no data-path collective.
  * local BA of one window: map points are partitioned over the ranks (`partition_observations`);
    the C library sums its per-iteration reduce buffer through the hook `make_allreduce_hook`
    installs (orbx_ba_set_allreduce).  Every rank gets identical poses, points and error values.
"""
import numpy as np


def shard_streams(n_streams, rank, world):
    """Streams (or frames of one stream) owned by `rank`: round robin, SURVEY §8e row 1."""
    return list(range(rank, n_streams, world))


def point_owner(mp_idx, world):
    """Owner rank of every map point index: round robin keeps the per-rank observation counts even
    for windows whose points have similar track lengths."""
    return np.asarray(mp_idx) % world


def partition_observations(obs, rank, world):
    """The observations of the points `rank` owns — ALL observations of each owned point, so that
    V_j, W_j and the point's Schur term are complete on one rank."""
    return obs[point_owner(obs["mp_idx"], world) == rank]


class _DeviceDoubles:
    """Zero-copy view of `n` f64 at a raw device pointer for torch.as_tensor."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)


def device_tensor(ptr, n, device):
    import torch
    return torch.as_tensor(_DeviceDoubles(ptr, n), device=device)


def make_allreduce_hook(device, group=None):
    """fn(dev_ptr, n_doubles, hip_stream) for Handle.set_allreduce: in-place SUM over the process
    group, ordered on the library's stream (the collective is enqueued with that stream current, so
    it waits for the kernels before it and the kernels after it wait for the collective)."""
    import torch
    import torch.distributed as dist

    def hook(ptr, n, stream):
        t = device_tensor(ptr, n, device)
        if stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=device)):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return hook


def init_native_rccl(handle, rank, world, group=None):
    """Gives `handle` its own RCCL communicator over all ranks (the library then calls ncclAllReduce itself, on its own
    stream — no Python in the LM loop).  Rank 0 draws the unique id through the library; it travels to the other ranks as
    an object broadcast over whatever torch.distributed group is up (gloo or nccl: only 128 bytes of bootstrap).  Collective:
    every rank calls it, before the first partitioned solve."""
    import torch.distributed as dist
    box = [handle.rccl_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    handle.init_rccl(box[0], rank, world)


def ba_solve_partitioned(handle, camera, cfg, poses_cw, fixed_cw, points, obs, rank, world, hook=None,
                         should_stop=None):
    """Point-partitioned solve_visual_ba: call on every rank with the same problem.  hook = None: the handle's native RCCL
    communicator (init_native_rccl / Handle.set_rccl_comm) carries the two all-reduces per iteration; otherwise the given
    hook (any transport: the CPU tests run gloo).  should_stop may answer differently on different ranks: the stop votes
    are all-reduced with the data, one rank asking stops every rank before the same iteration."""
    if hook is None:
        # No transport = every rank would silently solve its own partition as if it were the whole problem: refuse.
        if world > 1 and not (handle.has_collective() & 1):
            raise RuntimeError("ba_solve_partitioned: world = %d but the handle has no RCCL communicator (call init_native_rccl / "
                               "Handle.set_rccl_comm on every rank first) and no hook was given" % world)
        return handle.ba_solve_visual(camera, cfg, poses_cw, fixed_cw, points,
                                      partition_observations(obs, rank, world), should_stop)
    handle.set_allreduce(hook)
    try:
        return handle.ba_solve_visual(camera, cfg, poses_cw, fixed_cw, points,
                                      partition_observations(obs, rank, world), should_stop)
    finally:
        handle.set_allreduce(None)


def allreduce_max_seconds(seconds, device=None):
    """bench.py: max-over-ranks of a wall-clock interval."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
