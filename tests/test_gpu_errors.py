"""What the C-ABI does with calls it cannot serve -- and that a context which refused a call is as good as before.  The reference asserts or dereferences in these
places; a library behind a foreign-function boundary returns a status and a message (include/ht_mi355x.h: HT_ERR_ARG / HT_ERR_STATE, ht_last_error)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FR = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))


def test_refused_calls_leave_the_context_usable():
    from hand_tracking_samples_amd import native
    B = 8
    depth = FR["depth"][:B].reshape(B, -1); cams = FR["cam"][:B]; start = FR["startpose"][:B]
    ctx = native.Context(ol.MODEL, B)
    try:
        with pytest.raises(native.HTError, match="weights not loaded"):
            ctx.update_sync(depth, cams)                                   # no net yet
        ctx.load_weights(W.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(start)
        good = ctx.update_sync(depth, cams)
        ctx.tracker_reset(start)
        L, h = ctx.L, ctx.h
        u16 = C.POINTER(C.c_uint16); fp = C.POINTER(C.c_float)
        poses = np.zeros((B + 1, ctx.nb, 7), np.float32)
        big = np.zeros((B + 1, 4096), np.uint16); bigc = np.zeros((B + 1, 12), np.float32)
        dptr = lambda a: a.ctypes.data_as(u16)
        fptr = lambda a: a.ctypes.data_as(fp)
        assert L.ht_update_sync(h, dptr(big), fptr(bigc), B + 1, fptr(poses), None) != 0          # more frames than ht_create reserved
        assert b"capacity" in L.ht_last_error(h)
        assert L.ht_update_sync(h, dptr(big), fptr(bigc), 0, fptr(poses), None) != 0              # an empty batch is not a batch
        assert L.ht_update_sync(h, None, fptr(bigc), B, fptr(poses), None) != 0                   # null pointers
        assert L.ht_update_sync(h, dptr(big), None, B, fptr(poses), None) != 0
        assert L.ht_update_sync(h, dptr(big), fptr(bigc), B, None, None) != 0
        assert L.ht_tracker_reset(h, 4, B, fptr(poses)) != 0                                       # a slot range past the end
        assert L.ht_tracker_reset(h, -1, 2, fptr(poses)) != 0
        assert L.ht_get_state(h, 2, 0, B, fptr(np.zeros((B, ctx.nb, 13), np.float32))) != 0       # there are two models: 0 and 1
        frame = np.zeros((B, 66, 66), np.uint16)                                                    # a frame size the segmentation does not take (not a multiple of 4)
        assert L.ht_update_frames_sync(h, dptr(frame), fptr(bigc), 66, 66, C.c_float(0.17), B, fptr(poses), None) != 0
        assert b"frame size" in L.ht_last_error(h)
        with pytest.raises(native.HTError, match="128x128 net not loaded"):
            ctx.update_direct_sync(np.zeros((B, 128, 128), np.uint16), cams, 128)                   # the other net has no weights
        d = torch.zeros(B * 128 * 128 + 8, dtype=torch.int16, device="cuda")
        out = torch.zeros(B * ctx.nb * 7, dtype=torch.float32, device="cuda"); dc = torch.zeros(B * 12, dtype=torch.float32, device="cuda")
        ctx.load_weights128(W.make_cnnb128())
        with pytest.raises(native.HTError, match="16-byte aligned"):
            ctx.update_direct_dev(d.data_ptr() + 2, dc.data_ptr(), 128, 0, B, out.data_ptr(), 0)  # a device pointer the 128-bit loads cannot take
        with pytest.raises(AttributeError):
            ctx.set_params(no_such_parameter=1)
        # nothing above touched the trackers: the same update gives the same poses
        again = ctx.update_sync(depth, cams)
        assert np.array_equal(good, again)
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()


def test_a_context_without_a_model_serves_the_net_only():
    """CNN PoseInitializerCNN(std::string) on its own (handtrack.h:103-130): ht_create(NULL, ...)"""
    from hand_tracking_samples_amd import native
    ctx = native.Context(None, 4)
    try:
        ctx.load_weights(W.make_cnnb())
        x = np.random.default_rng(3).random((4, 4096), dtype=np.float32)
        y = ctx.cnn_eval(x)
        assert y.shape == (4, 2304) and np.isfinite(y).all()
        with pytest.raises(native.HTError, match="without a hand model"):
            ctx.update_sync(FR["depth"][:4].reshape(4, -1), FR["cam"][:4])
    finally:
        ctx.close()
