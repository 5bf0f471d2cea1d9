"""tests/parity_rule.py on the committed yardsticks (no GPU): what the rule allows is derived from the reference's own builds, and these are the numbers it derives."""
import os

import numpy as np

import parity_rule as pr

HERE = os.path.dirname(os.path.abspath(__file__))


def _spread(name):
    return np.load(os.path.join(HERE, "golden", name))


def test_frame_rule_accepts_the_tight_band_and_scaled_moves_only():
    sp = np.array([1e-6, 1e-4, 1e-4, 1e-6], np.float32); sq = np.array([1e-5, 1e-3, 1e-3, 1e-5], np.float32)
    dp = np.array([1e-5, 1.9e-4, 2.1e-4, 1e-4], np.float32); dq = np.array([1e-4, 1e-3, 1e-3, 1e-4], np.float32)
    ok, tight = pr.frame_rule(dp, dq, sp, sq)
    assert tight.tolist() == [True, False, False, False]
    assert ok.tolist() == [True, True, False, False]      # inside the band | outside where the reference is, within twice its move | beyond twice | outside where the reference is inside


def test_a_nan_fails_the_rule():
    ok, tight = pr.frame_rule(np.array([np.nan], np.float32), np.array([0.0], np.float32), np.array([1.0], np.float32), np.array([1.0], np.float32))
    assert not ok[0] and not tight[0]


def test_quaternion_distance_takes_one_sign_per_quaternion():
    a = np.zeros((1, 2, 7), np.float32); a[0, :, 6] = 1.0
    b = a.copy(); b[0, 1, 3:] *= -1.0
    dp, dq = pr.pose_diff(a, b)
    assert dp[0] == 0.0 and dq[0] == 0.0


def test_allow_lists_come_out_of_the_references_own_builds():
    """how many frames one FMA build of the reference fails when it is held against the other's yardstick the way the device is held (the smaller direction)"""
    assert pr.cross_build_failures(_spread("ref_spread1024.npz"), "other", 4.0, pr.CAP_TAKE_CNN) == 16
    assert pr.cross_build_failures(_spread("ref_spread1024_takecnn.npz"), "user", 4.0, pr.CAP_TAKE_CNN) == 36
    assert pr.cross_build_failures(_spread("ref_spread1024.npz"), "user", 2.0, pr.CAP) == 3
    sub = np.arange(64) * 4
    assert pr.cross_build_failures(_spread("ref_spread5e2e_256.npz"), "user", 2.0, pr.CAP_TAKE_CNN, sub) == 2
