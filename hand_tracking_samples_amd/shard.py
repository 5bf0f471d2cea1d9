"""Frame sharding across the GPUs of one node.

Frames are independent, so the batch is cut into one contiguous shard per rank and no collective touches the
data path; the only exchange is one all-gather of the per-frame poses (17 x 7 floats = 476 B per frame) at the end of a step,
which over xGMI is latency- not link-bound (3.9 MB per GPU at 8192 frames).  torch.distributed supplies the process group
(backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).
"""


def shard_range(n_frames, rank, world):
    """Contiguous [lo, hi) of rank's shard; the first n_frames % world ranks take one extra frame."""
    base, extra = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_poses(local_poses, world, out=None, force=False, async_op=False):
    """All-gather per-rank pose tensors [n_r, nb, 7] (equal n_r on every rank) into [world * n_r, nb, 7].
    `force`: run the collective even for a single rank (rehearsal of the RCCL path on a one-GPU box).
    `async_op`: do not make the caller's stream wait for the collective; returns (out, work) and the caller waits on `work` (or synchronises the
    device) before it reads `out` or overwrites `local_poses` -- this is how a step's exchange runs beside the next step's kernels."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return (local_poses, None) if async_op else local_poses
    if out is None:
        out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype, device=local_poses.device)
    work = dist.all_gather_into_tensor(out, local_poses.contiguous(), async_op=async_op)
    return (out, work) if async_op else out


def gather_poses_ragged(local_poses, counts):
    """All-gather shards of unequal length: pads to the largest shard, gathers once, then drops the padding."""
    import torch
    import torch.distributed as dist
    world = len(counts)
    if world == 1:
        return local_poses
    m = max(counts)
    pad = torch.zeros((m,) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype, device=local_poses.device)
    pad[: local_poses.shape[0]] = local_poses
    out = torch.empty((world * m,) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)])
