"""bench.py's rank logic with two ranks on gloo (CPU): which frames a rank takes, how the ranks agree on the exchange (the library's RCCL communicator or
torch.distributed) without anybody being left waiting inside ncclCommInitRank, and the book-keeping of the two pose-buffer pairs.  The functions under test are the
ones bench.py calls (hand_tracking_samples_amd/shard.py); the communicator itself is a stub here, the real one is rehearsed on the GPU box (tests/test_gpu_comm.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hand_tracking_samples_amd.shard import PoseBuffers, negotiate_library_gather, rank_frames


def test_rank_frames_offsets():
    # one GPU, 1024 frames: every distinct frame once, in order
    assert np.array_equal(rank_frames(1024, 0, 1, 1024), np.arange(1024))
    # BASELINE configs[3]: 8 ranks x 8192 frames -- every rank sees every distinct frame 8 times, and no two ranks start at the same frame
    firsts = set()
    for r in range(8):
        idx = rank_frames(8192, r, 8, 1024)
        assert len(idx) == 8192 and np.array_equal(np.bincount(idx, minlength=1024), np.full(1024, 8))
        firsts.add(int(idx[0]))
    assert len(firsts) == 8
    # the 64-frame set of configs[4]
    assert np.array_equal(np.bincount(rank_frames(1024, 3, 4, 64), minlength=64), np.full(64, 16))


def test_pose_buffers_wait_before_reuse():
    log = []
    b = PoseBuffers(lambda k, h: log.append(("wait", k, h)))
    for step in range(5):
        k = b.next_slot()
        assert k == step & 1
        log.append(("write", k))
        b.issued(k, "gather%d" % step)
    b.drain()
    # the gather of step s is waited for right before step s + 2 writes the same pair, never earlier; the last two at the end
    assert log == [("write", 0), ("write", 1), ("wait", 0, "gather0"), ("write", 0), ("wait", 1, "gather1"), ("write", 1), ("wait", 0, "gather2"), ("write", 0), ("wait", 0, "gather4"), ("wait", 1, "gather3")]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, scenario, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def make_uid():
            calls.append("uid")
            if scenario == "no_uid":
                raise RuntimeError("ncclGetUniqueId failed")
            return bytes(range(128))

        def join(uid):
            calls.append("join")
            assert uid == bytes(range(128))      # the id made on rank 0 arrived intact
            if scenario == "join_fails_on_1" and rank == 1:
                raise RuntimeError("ncclCommInitRank: invalid argument")
            if scenario == "all_good":
                dist.barrier()      # stands in for ncclCommInitRank, which returns only once every rank has entered it

        available = not (scenario == "rank1_has_no_rccl" and rank == 1)
        use, why = negotiate_library_gather(dist, torch.device("cpu"), rank, world, available, make_uid, join)
        q.put((rank, scenario, use, why, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["all_good", "rank1_has_no_rccl", "join_fails_on_1", "no_uid"])
def test_two_ranks_agree_on_the_exchange(scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)      # a rank left waiting inside the stub would time out here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, _, use0, why0, calls0), (r1, _, use1, why1, calls1) = res
    assert (r0, r1) == (0, 1) and use0 == use1      # both ranks take the same exchange, whatever happened
    if scenario == "all_good":
        assert use0 and why0 is None and calls0 == ["uid", "join"] and calls1 == ["join"]
    elif scenario == "rank1_has_no_rccl":
        assert not use0 and "every rank" in why0 and calls0 == [] and calls1 == []      # nobody even made an id, let alone entered the communicator
    elif scenario == "join_fails_on_1":
        assert not use0 and why0 == "another rank failed" and "invalid argument" in why1
    else:
        assert not use0 and "unique id" in why0 and calls0 == ["uid"] and calls1 == []
