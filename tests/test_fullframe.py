"""HandTracker::update on frames that are not 64x64 (handtrack.h:693-785): the tracker segments the frame for the CNN (HandSegmentVR with
segment_scale), takes the point cloud and FitError from the full-resolution frame, and hands segment.cam.pose to the pose-driven stages.

  fullframe5.htfx    BASELINE configs[4]: 128x128 frames, the 26-bone hand (tests/golden/make_model_hand26.py), 4 animation-bank rows
  fullframe320.htfx  the application's native 320x240 camera (synthetic-tracker.cpp:98), the 17-bone hand, 2 rows, ~1600 points per frame
Both from the reference (`ref_harness fullframe`), two consecutive updates per frame."""
import ctypes as C
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "config5": (os.path.join(HERE, "golden", "fullframe5.htfx"), os.path.join(HERE, "golden", "model_hand26.htfx"), 26),
    "qvga": (os.path.join(HERE, "golden", "fullframe320.htfx"), ol.MODEL, 17),
}
GOLD = {k: htfx.load(v[0]) for k, v in CASES.items()}
FRAMES = [(k, f) for k in CASES for f in range(len(GOLD[k]["rows"]))]


@pytest.fixture(scope="module")
def weights():
    return W.make_cnnb()


@pytest.mark.parametrize("case,f", FRAMES)
def test_oracle_full_frame_update_matches_reference(weights, case, f):
    G, (_, model, nb) = GOLD[case], CASES[case]
    w, h = (int(x) for x in G["dims"])
    pre = "f%d/" % f
    orc = ol.Oracle(weights, model=model)
    try:
        orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
        orc.reset(G[pre + "startpose"])
        user = np.zeros((nb, 7), np.float32)
        cam = ol.camera(G[pre + "cam"], w, h)
        depth = np.ascontiguousarray(G[pre + "depth"])
        orc.L.ho_update(orc.h, ol.u16ptr(depth), C.byref(cam), ol.fptr(user))
        assert np.array_equal(orc.get_state(1), G[pre + "uw_other_after_cnn"])
        assert np.array_equal(orc.get_state(0), G[pre + "uw_hand_pass2"])
        assert np.array_equal(user, G[pre + "uw_pose_user"])
        assert orc.flags() == (G[pre + "uw_final"][0], int(G[pre + "uw_final"][1]), int(G[pre + "uw_final"][2]))
        orc.L.ho_update(orc.h, ol.u16ptr(depth), C.byref(cam), ol.fptr(user))
        assert np.array_equal(user, G[pre + "uw2_pose_user"])
        assert np.array_equal(orc.get_state(0), G[pre + "uw2_hand"])
    finally:
        orc.close()
