"""BASELINE configs[4] ("config 5"): 128x128 depth frames of a 26-bone hand (tests/golden/make_model_hand26.py) through the
application's sequence -- HandSegmentVR, poses re-based into the segment camera, then the tracker's unit of work on the tile
(synthetic-tracker.cpp:204-215, handtrack.h:693-785).  tests/golden/config5.htfx comes from the reference running that very model
(`HT_REF_MODEL_JSON=... ref_harness config5`), 4 animation-bank rows."""
import ctypes as C
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL26 = os.path.join(HERE, "golden", "model_hand26.htfx")
G = htfx.load(os.path.join(HERE, "golden", "config5.htfx"))
NF = len(G["rows"])


@pytest.fixture(scope="module")
def weights():
    assert float(np.float32(W.DEFAULT_SEED)) == G["weights_seed_gain"][0] and G["weights_seed_gain"][1] == W.DEFAULT_FC2_GAIN
    return W.make_cnnb()


@pytest.mark.parametrize("f", range(NF))
def test_oracle_segments_the_128_frame_like_the_reference(f):
    pre = "f%d/" % f
    tile, cam, _, _ = ol.segment_vr(G[pre + "depth128"], G[pre + "cam128"], 0xF, (0.1, 0.7), 0.17)
    assert np.array_equal(tile, G[pre + "tile"])
    assert np.array_equal(cam, G[pre + "segcam"])


@pytest.mark.parametrize("f", range(NF))
def test_oracle_tracks_the_26_bone_hand_like_the_reference(weights, f):
    pre = "f%d/" % f
    orc = ol.Oracle(weights, model=MODEL26)
    try:
        assert orc.nb == 26
        orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
        orc.reset(G[pre + "startpose"])
        user = np.zeros((26, 7), np.float32)
        cam = ol.camera(G[pre + "cam"])
        orc.L.ho_update(orc.h, ol.u16ptr(G[pre + "tile"]), C.byref(cam), ol.fptr(user))
        assert np.array_equal(orc.get_state(1), G[pre + "uw_other_final"])
        assert np.array_equal(orc.get_state(0), G[pre + "uw_hand_pass2"])
        assert np.array_equal(user, G[pre + "uw_pose_user"])
    finally:
        orc.close()


# tolerances as tests/test_gpu_solver.py states them: same arithmetic as the reference when the hand model never sees the CNN pose,
# MFMA-rounding of the CNN amplified by the hard-driven MultiStepSim when the CNN pose is accepted
POS_TOL, QUAT_TOL, FULL_POS_TOL, FULL_QUAT_TOL = 2e-5, 2e-4, 2e-4, 2e-3


@pytest.mark.gpu
def test_gpu_config5_matches_reference(weights):
    """Device path: k_segment on the 128x128 frames, then the whole unit of work with the 26-bone model (325 body pairs, 25 joints)."""
    from hand_tracking_samples_amd import native
    ctx = native.Context(MODEL26, NF)
    try:
        assert (ctx.nb, ctx.nj) == (26, 25)
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        depth = np.stack([G["f%d/depth128" % f] for f in range(NF)]); cams = np.stack([G["f%d/cam128" % f] for f in range(NF)])
        tiles, segcams = ctx.segment_vr(depth, cams, 0xF, (0.1, 0.7), 0.17)
        for f in range(NF):
            assert np.array_equal(tiles[f], G["f%d/tile" % f])
            assert np.abs(segcams[f] - G["f%d/segcam" % f]).max() <= 2e-7 * max(1.0, np.abs(G["f%d/segcam" % f]).max())
        # the application re-bases the poses into the segment camera and clears its pose (synthetic-tracker.cpp:207-215); the golden holds both
        ctx.tracker_reset(np.stack([G["f%d/startpose" % f] for f in range(NF)]))
        poses, cnn = ctx.update_sync(np.stack([G["f%d/tile" % f] for f in range(NF)]), np.stack([G["f%d/cam" % f] for f in range(NF)]), want_cnn=True)
        assert np.abs(cnn - np.stack([G["f%d/cnn_output" % f] for f in range(NF)])).max() <= 2e-5
        for f in range(NF):
            ref = G["f%d/uw_pose_user" % f]
            accepted = G["f%d/uw_accept" % f][0] > 0
            dp = np.abs(poses[f, :, :3] - ref[:, :3]).max(); dq = np.abs(poses[f, :, 3:] - ref[:, 3:]).max()
            print("config5 frame %d (cnn pose %s, %d contacts, %d points): |dpos| %.2e |dquat| %.2e" % (f, "accepted" if accepted else "rejected", G["f%d/ncontacts_final" % f][0], G["f%d/ncontacts_final" % f][1], dp, dq))
            assert dp <= (FULL_POS_TOL if accepted else POS_TOL) and dq <= (FULL_QUAT_TOL if accepted else QUAT_TOL)
    finally:
        ctx.close()
