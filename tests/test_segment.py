"""HandSegmentVR (handtrack.h:280-344), the step before the tracker for full-size frames.

tests/golden/segment6.npz holds six 320x240 frames rendered from the reference's animation bank and what the reference's own
HandSegmentVR returns for them (tile + segment camera), plus its two intermediate images, for the entry options the reference
supports (0xF as the apps call it; 1, 2, 4, 8 and 5 on the first two frames).
  * CPU: the C restatement (oracle/ho_segment.c) must reproduce every array bit for bit.
  * GPU: ht_segment_vr must reproduce the tiles exactly and the cameras to float rounding (the rotation uses sin/cos/atan2 whose
    last bit may differ between glibc and the device library), and agree with the restatement on a larger seeded batch.
"""
import os

import numpy as np
import pytest

import oracle_lib
from hand_tracking_samples_amd import native

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "segment6.npz"))
CASES = [(i, o) for i in range(6) for o in ([15, 1, 2, 4, 8, 5] if i < 2 else [15])]
WRANGE = (0.1, 0.70)      # as synthetic-tracker.cpp / train-cnn.cpp call it


@pytest.mark.parametrize("frame,opt", CASES)
def test_oracle_matches_reference(frame, opt):
    tile, cam, small, dt = oracle_lib.segment_vr(FIX["s%d__depth" % frame], FIX["s%d__cam" % frame], opt, WRANGE)
    assert np.array_equal(small, FIX["s%d__small" % frame])
    assert np.array_equal(dt.astype(np.int32), FIX["s%d__dt" % frame])
    assert np.array_equal(tile, FIX["s%d__o%d__tile" % (frame, opt)])
    assert cam.tobytes() == FIX["s%d__o%d__cam" % (frame, opt)].tobytes()


def test_oracle_passes_64x64_through():
    d = (np.arange(4096, dtype=np.uint16) * 7 % 5000).reshape(64, 64)
    cam = np.arange(12, dtype=np.float32)
    tile, co, _, _ = oracle_lib.segment_vr(d, cam)
    assert np.array_equal(tile, d) and np.array_equal(co, cam)


def synth_frames(n, seed=3):
    """Blobs entering from an edge: a tapered arm from a random border point to a disc-shaped hand, at varying depth."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:240, 0:320].astype(np.float32)
    out = np.full((n, 240, 320), 4000, np.uint16)
    for k in range(n):
        side = rng.randint(4)
        e = np.array([rng.uniform(40, 280), 239.0]) if side == 0 else np.array([rng.uniform(40, 280), 0.0]) if side == 1 else \
            np.array([319.0, rng.uniform(30, 210)]) if side == 2 else np.array([0.0, rng.uniform(30, 210)])
        c = np.array([rng.uniform(90, 230), rng.uniform(70, 170)])
        z = rng.uniform(250, 640)
        r = rng.uniform(22, 48)
        hand = (xx - c[0]) ** 2 + (yy - c[1]) ** 2 < r * r
        t = np.clip(((xx - e[0]) * (c[0] - e[0]) + (yy - e[1]) * (c[1] - e[1])) / max(1e-3, ((c - e) ** 2).sum()), 0, 1)
        arm = (xx - (e[0] + t * (c[0] - e[0]))) ** 2 + (yy - (e[1] + t * (c[1] - e[1]))) ** 2 < (0.45 * r) ** 2
        depth = z + 0.15 * (xx - c[0]) + 0.1 * (yy - c[1]) + rng.uniform(-3, 3, size=xx.shape)
        m = hand | arm
        out[k][m] = np.clip(depth[m], 120, 3000).astype(np.uint16)
    cams = np.tile(np.array([305, 305, 160, 120, 0.001, 0, 0, 0, 0, 0, 0, 1], np.float32), (n, 1))
    return out, cams


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(os.path.join(HERE, "golden", "model_hand17.htfx"), max_batch=1)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_matches_reference_fixture(ctx):
    for opt in (15, 1, 2, 4, 8, 5):
        frames = [i for i, o in CASES if o == opt]
        depth = np.stack([FIX["s%d__depth" % i] for i in frames]); cams = np.stack([FIX["s%d__cam" % i] for i in frames])
        tiles, co = ctx.segment_vr(depth, cams, opt, WRANGE)
        for k, i in enumerate(frames):
            et, ec = FIX["s%d__o%d__tile" % (i, opt)], FIX["s%d__o%d__cam" % (i, opt)]
            assert np.abs(co[k] - ec).max() <= 2e-7, (i, opt, co[k] - ec)
            assert (tiles[k] != et).sum() <= (0 if co[k].tobytes() == ec.tobytes() else 8), (i, opt, (tiles[k] != et).sum())


@pytest.mark.gpu
def test_gpu_matches_restatement_on_a_batch(ctx):
    depth, cams = synth_frames(48)
    tiles, co = ctx.segment_vr(depth, cams, 0xF, (0.1, 0.65))
    exact = 0
    for k in range(len(depth)):
        et, ec, _, _ = oracle_lib.segment_vr(depth[k], cams[k], 0xF, (0.1, 0.65))
        assert np.abs(co[k] - ec).max() <= 2e-7 * max(1.0, np.abs(ec).max()), (k, co[k] - ec)
        same = co[k].tobytes() == ec.tobytes()
        exact += same
        assert (tiles[k] != et).sum() <= (0 if same else 8), (k, (tiles[k] != et).sum())
    assert exact >= len(depth) * 3 // 4      # the rare 1-ulp differences of sin/cos/atan2 aside, cameras are identical


@pytest.mark.gpu
def test_gpu_passes_64x64_through_and_rejects_odd_sizes(ctx):
    d = (np.arange(2 * 4096, dtype=np.uint16) * 7 % 5000).reshape(2, 64, 64)
    cams = np.arange(24, dtype=np.float32).reshape(2, 12)
    tiles, co = ctx.segment_vr(d, cams)
    assert np.array_equal(tiles, d) and np.array_equal(co, cams)
    with pytest.raises(native.HTError):
        ctx.segment_vr(np.zeros((1, 241, 322), np.uint16), cams[:1])
    with pytest.raises(native.HTError):
        ctx.segment_vr(np.zeros((1, 480, 640), np.uint16), cams[:1])
