"""The result gather of the multi-GPU layout through the C-ABI (ht_comm_* / ht_gather_poses_dev: RCCL's ncclAllGather on the context's communication
stream), rehearsed with the one rank a GPU box has: the communicator is made from a unique id exactly as with N ranks, the gather runs behind the
update on its own stream, and what lands in the gathered array is this rank's poses.  (Sharding arithmetic for N > 1: tests/test_shard_gloo.py.)"""
import numpy as np
import pytest
import torch  # noqa: F401  -- at import (= collection) time on purpose: torch brings its own copies of the ROCm runtime and of RCCL, and a process in which OUR library has
#                  already initialised HIP when torch is first imported ends up with RCCL bound to a second, uninitialised runtime ("no ROCm-capable device is detected"
#                  from ncclCommInitRank).  A Python host that uses both imports torch first (bench.py does); a C++ host has only the system's copies.

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_gather_through_the_library_on_one_rank(golden, weights):
    import torch
    from hand_tracking_samples_amd import native
    nf = 8
    depth = np.stack([golden["f%d/depth" % f].reshape(-1) for f in range(nf)]); cams = np.stack([golden["f%d/cam" % f] for f in range(nf)])
    start = np.stack([golden["f%d/startpose" % f] for f in range(nf)])
    dev = torch.device("cuda", 0)
    ctx = native.Context(ol.MODEL, nf)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        uid = native.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        ctx.comm_init(1, 0, uid)
        assert ctx.comm_info() == (1, 0)
        with pytest.raises(native.HTError, match="already has a communicator"):
            ctx.comm_init(1, 0, uid)
        d_depth = torch.from_numpy(depth.view(np.int16)).to(dev); d_cams = torch.from_numpy(cams).to(dev); d_start = torch.from_numpy(start).to(dev)
        local = [torch.zeros((nf, 17, 7), dtype=torch.float32, device=dev) for _ in range(2)]
        allp = [torch.full((nf, 17, 7), -1.0, dtype=torch.float32, device=dev) for _ in range(2)]
        s = torch.cuda.current_stream(dev).cuda_stream
        for step in range(4):      # two buffer pairs in turn, as bench.py drives them
            k = step & 1
            ctx.gather_wait(k, s)
            ctx.update_dev(d_depth.data_ptr(), d_cams.data_ptr(), d_start.data_ptr(), nf, local[k].data_ptr(), s)
            ctx.gather_poses_dev(local[k].data_ptr(), allp[k].data_ptr(), nf, k, s)
        ctx.gather_wait_host(0); ctx.gather_wait_host(1)
        torch.cuda.synchronize()
        for k in range(2):
            assert torch.equal(local[k], allp[k])
        ref = np.stack([golden["f%d/uw_pose_user" % f] for f in range(nf)])
        got = allp[1].cpu().numpy()
        assert np.abs(got[:, :, :3] - ref[:, :, :3]).max() <= 2e-4 and np.abs(got[:, :, 3:] - ref[:, :, 3:]).max() <= 2e-3
    finally:
        ctx.close()
