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


def rank_frames(frames_per_gpu, rank, world, n_distinct, rank_offset=131):
    """Which distinct frame every slot of a rank's shard carries (bench.py): the global list walks the n_distinct frames round and round, rank r takes its contiguous
    shard [r * frames_per_gpu, (r + 1) * frames_per_gpu) of it, started rank_offset * r frames further on, so that ranks do not work on identical batches when
    frames_per_gpu is a multiple of n_distinct ("same generators, different animbank offsets")."""
    import numpy as np
    lo, hi = shard_range(frames_per_gpu * world, rank, world)
    first = (lo + rank_offset * rank) % n_distinct
    return (first + np.arange(hi - lo)) % n_distinct


def negotiate_library_gather(dist, device, rank, world, available, make_unique_id, join):
    """The ranks of a job agree on whether the exchange runs through the library's own RCCL communicator (ht_comm_init / ht_gather_poses_dev) or through
    torch.distributed.  Every rank goes through the SAME collectives whatever fails locally, and nobody enters ncclCommInitRank -- which blocks until all
    ranks have arrived -- before all ranks have said they can load RCCL:
      1. MIN over `available` (ht_comm_available on every rank);
      2. rank 0 makes the 128-byte id (make_unique_id() -> bytes), byte 128 says whether it could; broadcast;
      3. every rank joins (join(id_bytes); an exception = this rank could not);
      4. MIN over who joined.
    Returns (use_library, why_not): why_not is None when the library's gather is used by everybody.  `dist` is torch.distributed (any backend: gloo in the CPU tests)."""
    import torch
    avail = torch.tensor([1 if available else 0], dtype=torch.int32, device=device)
    dist.all_reduce(avail, op=dist.ReduceOp.MIN)
    all_have = int(avail.item()) == 1
    uid = torch.zeros(129, dtype=torch.uint8, device=device)
    if rank == 0 and all_have:
        try:
            raw = bytes(make_unique_id())
            assert len(raw) == 128
            uid[:128].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            uid[128] = 1
        except Exception as e:      # noqa: BLE001 -- whatever went wrong, the other ranks must still be told
            import sys
            sys.stderr.write("rank 0: RCCL unique id unavailable (%s)\n" % e)
    dist.broadcast(uid, 0)
    joined, why = 0, ("rank 0 could not make an RCCL unique id" if all_have else "RCCL cannot be loaded on every rank")
    if int(uid[128].item()) == 1:
        try:
            join(bytes(uid[:128].cpu().numpy().tobytes()))
            joined = 1
        except Exception as e:      # noqa: BLE001
            import sys
            why = str(e)
            sys.stderr.write("rank %d: ht_comm_init failed (%s)\n" % (rank, e))
    ok = torch.tensor([joined], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return True, None
    return False, (why if not joined else "another rank failed")


class PoseBuffers:
    """The two pose-buffer pairs of a rank (bench.py): step k writes pair k & 1 and issues its exchange; before a pair is written again
    the exchange of two steps ago that still reads it has to be through.  wait(k) / issued(k, handle) keep that book; `waiter` is what waits for a handle
    (ht_gather_wait for the library's gather, work.wait() for torch.distributed's)."""

    def __init__(self, waiter):
        self.pending = [None, None]
        self.waiter = waiter
        self.step = 0

    def next_slot(self):
        k = self.step & 1
        self.step += 1
        if self.pending[k] is not None:
            self.waiter(k, self.pending[k])
            self.pending[k] = None
        return k

    def issued(self, k, handle):
        self.pending[k] = handle

    def drain(self):
        for k in (0, 1):
            if self.pending[k] is not None:
                self.waiter(k, self.pending[k])
                self.pending[k] = None
