"""ht_update_dev takes any pointer the device can read: a PINNED host allocation (hipHostMalloc: one address for host and device) for the depth frames and for the pose output works
as it is -- k_prepare reads the frames over the host link, the update's last solve writes the poses to host memory -- and gives the poses of the device-resident call bit for bit.
bench.py's host_io leg measures what that is worth against uploads on a copy stream."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FR = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))


def test_pinned_host_frames_and_poses_equal_the_resident_call():
    from hand_tracking_samples_amd import native
    B = 256
    dev = torch.device("cuda", 0)
    depth = FR["depth"][:B].reshape(B, -1).astype(np.uint16); cams = FR["cam"][:B].astype(np.float32); start = FR["startpose"][:B].astype(np.float32)
    ctx = native.Context(ol.MODEL, B)
    try:
        ctx.load_weights(W.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
        s = torch.cuda.current_stream(dev)
        d_depth = torch.from_numpy(depth.view(np.int16)).to(dev); d_cams = torch.from_numpy(cams).to(dev); d_start = torch.from_numpy(start).to(dev)
        d_out = torch.empty((B, ctx.nb, 7), dtype=torch.float32, device=dev)
        h_depth = torch.from_numpy(depth.view(np.int16)).pin_memory(); h_out = torch.zeros((B, ctx.nb, 7), dtype=torch.float32).pin_memory()
        res = []
        for u in range(2):      # two updates: the second on the carried state
            ctx.update_dev(d_depth.data_ptr(), d_cams.data_ptr(), d_start.data_ptr() if u == 0 else 0, B, d_out.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize(); res.append(d_out.cpu().numpy().copy())
        for u in range(2):
            ctx.update_dev(h_depth.data_ptr(), d_cams.data_ptr(), d_start.data_ptr() if u == 0 else 0, B, h_out.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(h_out.numpy(), res[u]), "update %d" % u
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()
