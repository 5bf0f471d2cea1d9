"""Whole unit of work (CNN + decode + MultiStepSim + accept + 3 FitPointCloud passes) on the bench's own 1024 distinct frames, device against the
REFERENCE frame by frame: tests/golden/poses1024.htfx holds what the reference's own code (IEEE build, oracle/ref_harness.cpp `poses`) returns for each of
them; the C restatement reproduces them bit for bit (test_oracle_vs_golden.py on the first 256).  The frames cover the data-dependent branches (chamber
on/off at 400 points, full reset, CNN pose accepted / rejected).

Tolerance, and what it is held against.  tests/test_gpu_exact_solver.py shows that the device path IS the reference's computation once its solver sweeps
are the reference's own: bit for bit on every frame.  The product's sweeps apply the same rows in the same order in Jacobian form with fused multiply-adds
(csrc/ht_quad.hpp): one more floating-point BUILD of the algorithm, like the reference compiled with FMA contraction.  How far a frame moves under such a
change is a property of the frame -- a few frames sit next to a discrete decision (closest bone of a point, a contact appearing) and amplify a rounding
difference a thousandfold, in every build.  tests/golden/ref_spread1024.npz holds, per frame, how far the reference's own two FMA-contracted builds move
from its IEEE build (tests/golden/ref_flag_spread.py; summary profiles/r04_reference_build_spread.json: 1017 / 1014 of 1024 frames within 2e-5 m / 2e-4,
1020 / 1019 within 2e-4 m / 2e-3, worst 4.1e-4 m / 9.2e-3).  The device has to do at least as well, frame by frame: a frame may leave the tight band only if
the reference's own builds leave it there too, and then by no more than twice what they do."""
import ctypes as C
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N = 256


import parity_rule as pr

OTHER_FACTOR = 4.0      # othermodel: hard-driven through MultiStepSim from heat-maps the device accumulates on MFMA tiles (the reference's FMA builds contract the net's sums too, differently)


def _diff(got, ref):
    return pr.pose_diff(got, ref)


def _run_batch(weights, take_cnn):
    from hand_tracking_samples_amd import native
    n = 1024
    d = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))
    depth, cams, start = d["depth"].reshape(n, -1), d["cam"], d["startpose"]
    ctx = native.Context(ol.MODEL, n)
    ctx.load_weights(weights)
    ctx.set_params(microforce=3.0, mainthreadpasses=3, always_take_cnn=1 if take_cnn else 0)
    ctx.tracker_reset(start)
    got, cnn = ctx.update_sync(depth, cams, want_cnn=True)
    other = ctx.get_state(1, n)[:, :, :7]
    pfe, ini = ctx.tracker_flags(n)
    assert ctx.capacity_events() == (0, 0, 0)
    ctx.close()
    return got, other, ini, cnn


def _restatement_given_heat_maps(weights, cnn, take_cnn):
    """the CPU restatement (pinned on the reference bit for bit) on the 1024 frames, given the DEVICE's heat-maps: what the reference's arithmetic makes of the very CNN
    output the device's solver worked from -- the net's own rounding (MFMA accumulation, <= 2.6e-6: tests/test_gpu_cnn.py) is taken out of the comparison"""
    d = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3; orc.head.par.always_take_cnn = 1 if take_cnn else 0
    n = len(cnn)
    user = np.zeros((n, 17, 7), np.float32); other = np.zeros((n, 17, 7), np.float32)
    try:
        for i in range(n):
            orc.reset(d["startpose"][i])
            cam = ol.camera(d["cam"][i])
            y = np.ascontiguousarray(cnn[i]); orc.L.ho_set_cnn_override(orc.h, ol.fptr(y))
            orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(d["depth"][i]).reshape(-1)), C.byref(cam), ol.fptr(user[i]))
            other[i] = orc.get_state(1)[:, :7]
        orc.L.ho_set_cnn_override(orc.h, None)
    finally:
        orc.close()
    return user, other


def _distribution(dev, ref, what):
    """the CNN-driven poses against the fixture: no frame-by-frame yardstick exists for them (the device's heat-maps differ from the reference's by the net's own rounding,
    which a hard-driven MultiStepSim amplifies frame by frame differently than it amplifies the reference's FMA contraction), so: as a distribution"""
    dd = np.maximum(*dev); sr = np.maximum(*ref)
    pd, ps = np.percentile(dd, [50, 90, 99]), np.percentile(sr, [50, 90, 99])
    nd, ns = int(((dev[0] > pr.TIGHT[0]) | (dev[1] > pr.TIGHT[1])).sum()), int(((ref[0] > pr.TIGHT[0]) | (ref[1] > pr.TIGHT[1])).sum())
    print("  %s against the fixture: device p50 %.1e p90 %.1e p99 %.1e max %.1e, %d frames outside 2e-5 m / 2e-4; the reference's FMA builds p50 %.1e p90 %.1e p99 %.1e max %.1e, %d frames" % (what, *pd, dd.max(), nd, *ps, sr.max(), ns))
    assert (pd <= 2 * ps).all() and nd <= ns and dev[0].max() <= pr.CAP_TAKE_CNN[0] and dev[1].max() <= pr.CAP_TAKE_CNN[1]


def _check_batch(weights, take_cnn, refp, spread, what):
    got, other, ini, cnn = _run_batch(weights, take_cnn)
    n = len(got)
    assert np.isfinite(got).all() and np.isfinite(other).all()
    # (1) what does NOT hang on the net frame by frame against the FIXTURE (tests/parity_rule.py): the user poses without always_take_cnn -- outside the tight band only where
    #     the reference's own rebuilds are and by no more than twice their move, nothing beyond the cap.  The CNN-driven poses (othermodel; with always_take_cnn the user poses
    #     too) against the fixture as a distribution.
    sp, sq = pr.spread_of(spread, "user")
    dp, dq = _diff(got, refp["uw_pose_user"])
    if not take_cnn:
        u = pr.summary(got, refp["uw_pose_user"], sp, sq)
        ok, tight = pr.frame_rule(dp, dq, sp, sq)
        print("%s vs reference, %d frames: exact %d, within 2e-5 m / 2e-4: %d (the reference's FMA builds: %d), within 2e-4 m / 2e-3: %d; |dpos| p50 %.1e p99 %.1e max %.2e m, |dquat| p50 %.1e p99 %.1e max %.2e"
              % (what, n, int(((dp == 0) & (dq == 0)).sum()), u["within_2e-5m_2e-4"], u["reference_fma_builds_within_2e-5m_2e-4"], u["within_2e-4m_2e-3"],
                 np.percentile(dp, 50), np.percentile(dp, 99), dp.max(), np.percentile(dq, 50), np.percentile(dq, 99), dq.max()))
        for i in np.nonzero(~tight)[0][:24]:
            print("  frame %4d leaves the tight band: device %.2e m / %.2e, the reference's own FMA builds %.2e m / %.2e (sensitivity rank %d of %d)%s" % (i, dp[i], dq[i], sp[i], sq[i], int((sq > sq[i]).sum()), n, "" if ok[i] else "   <-- FAILS the rule"))
        assert u["finite"] and not u["frames_failing_the_rule"] and u["ok"], u
    else:
        _distribution((dp, dq), (sp, sq), "user poses (CNN-driven: always_take_cnn)")
    _distribution(_diff(other, refp["other_pose"]), pr.spread_of(spread, "other"), "othermodel (CNN-driven)")
    # (2) the CNN-driven half with the net's own rounding taken out: against the restatement GIVEN THE DEVICE'S HEAT-MAPS the solver's rounding is the only difference.  A frame
    #     next to a discrete decision (a closest bone, a contact appearing in one of the five hard-driven steps) flips under one rounding perturbation and not under another:
    #     WHICH frames move is not a property two builds share (measured: of the frames the device moves out of the tight band here, a third are not among those the
    #     reference's own FMA builds move), HOW MANY and HOW FAR is.  So: no more frames outside the tight band than the reference's own rebuilds leave there, percentiles
    #     within twice theirs, the cap -- and every frame the per-frame rule would fail is printed with both moves.
    ru, ro = _restatement_given_heat_maps(weights, cnn, take_cnn)
    for name, dev, res, key in (("othermodel", other, ro, "other"), ("user poses", got, ru, "user")):
        sp2, sq2 = pr.spread_of(spread, key)
        d2 = _diff(dev, res)
        ok2, tight2 = pr.frame_rule(d2[0], d2[1], sp2, sq2, OTHER_FACTOR if (key == "other" or take_cnn) else 2.0, pr.CAP_TAKE_CNN)
        print("  %s against the restatement given the device's heat-maps: %d of %d within 2e-5 m / 2e-4 (the reference's FMA builds against its IEEE build: %d), max %.2e m / %.2e"
              % (name, int(tight2.sum()), n, int(((sp2 <= pr.TIGHT[0]) & (sq2 <= pr.TIGHT[1])).sum()), d2[0].max(), d2[1].max()))
        for i in np.nonzero(~ok2)[0][:40]:
            print("    frame %4d: device %.2e m / %.2e, the reference's own FMA builds %.2e m / %.2e" % (i, d2[0][i], d2[1][i], sp2[i], sq2[i]))
        if key == "user" and not take_cnn:
            assert ok2.all()      # the user poses without always_take_cnn: frame by frame here too
        else:
            # CNN-driven: the frames that fail the per-frame rule are counted against what the reference's own two FMA builds do to EACH OTHER under the same rule
            # (parity_rule.cross_build_failures).  That each of these frames is a rounding difference amplified and not a defect: tests/test_gpu_teacher_forced.py (every
            # single step of all 1024 frames inside 1e-6 m / 2e-5 from the restatement's state), profiles/r06_notes.md (their growth step by step).
            allowed = pr.cross_build_failures(spread, key, OTHER_FACTOR, pr.CAP_TAKE_CNN)
            print("    frames failing the per-frame rule: %d; the reference's own FMA builds held against each other: %d" % (int((~ok2).sum()), allowed))
            assert int((~ok2).sum()) <= allowed
        _distribution(d2, (sp2, sq2), name + " (solver rounding only)")
    # (3) the tracker's discrete state after the frame: `initializing` (handtrack.h:781) on every frame
    assert np.array_equal(ini, refp["flags"][:, 1].astype(np.int32))


def test_batch_against_reference(weights):
    _check_batch(weights, False, htfx.load(os.path.join(HERE, "golden", "poses1024.htfx")), np.load(os.path.join(HERE, "golden", "ref_spread1024.npz")), "always_take_cnn = 0")


def test_batch_against_reference_always_take_cnn(weights):
    """The application's always_take_cnn switch (synthetic-tracker.cpp:91,127,240; accept rule handtrack.h:720-722): EVERY frame's user pose now depends on the net, its
    decode, MultiStepSim and the accept step (without the switch the accept branch fires on ~3 % of these frames and the user poses pin the three main passes only)."""
    _check_batch(weights, True, htfx.load(os.path.join(HERE, "golden", "poses1024_takecnn.htfx")), np.load(os.path.join(HERE, "golden", "ref_spread1024_takecnn.npz")), "always_take_cnn = 1")


def test_batch_against_restatement(weights):
    """The same comparison against the C restatement run beside the device (it is pinned to the fixture above bit for bit on the build box; here it
    shows that the checker that travels to the GPU box agrees with the fixture on this host's libm too)."""
    from hand_tracking_samples_amd import native
    d = np.load(os.path.join(HERE, "golden", "frames256.npz"))
    idx = np.arange(64) * 4
    depth, cams, start = d["depth"][idx].reshape(64, -1), d["cam"][idx], d["startpose"][idx]
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    ref = np.zeros((64, 17, 7), np.float32)
    for k in range(64):
        orc.reset(start[k])
        cam = ol.camera(cams[k])
        orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[k])), C.byref(cam), ol.fptr(ref[k]))
    orc.close()
    assert np.array_equal(ref, htfx.load(os.path.join(HERE, "golden", "poses256.htfx"))["uw_pose_user"][idx])
    ctx = native.Context(ol.MODEL, 64)
    ctx.load_weights(weights)
    ctx.set_params(microforce=3.0, mainthreadpasses=3)
    ctx.tracker_reset(start)
    got = ctx.update_sync(depth, cams)
    ctx.close()
    dp, dq = _diff(got, ref)
    tight = (dp <= 2e-5) & (dq <= 2e-4)
    print("vs restatement, 64 frames: tight %d, worst |dpos| %.2e m |dquat| %.2e" % (int(tight.sum()), dp.max(), dq.max()))
    assert tight.sum() >= 62


def test_full_size_batch_properties(weights):
    """BASELINE configs[2] at its full size (1024 frames = the 256 bench frames four times): size-independent properties instead of a
    frame-by-frame oracle run.  (1) A frame's result does not depend on where it sits in the batch: the four copies of every frame agree
    bit for bit.  (2) The whole step is deterministic: a second run from the same start poses reproduces every pose bit for bit (atomics
    are only used where the order cannot matter).  (3) Every pose is finite with unit quaternions."""
    from hand_tracking_samples_amd import native
    B = 1024
    d = np.load(os.path.join(HERE, "golden", "frames256.npz"))
    idx = np.arange(B) % 256
    depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
    ctx = native.Context(ol.MODEL, B)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(start)
        a = ctx.update_sync(depth, cams)
        ctx.tracker_reset(start)
        b = ctx.update_sync(depth, cams)
        assert ctx.capacity_events() == (0, 0, 0)      # no result depends on the contact kernel's capacities
    finally:
        ctx.close()
    assert np.array_equal(a, b)
    for k in range(1, 4):
        assert np.array_equal(a[:256], a[256 * k:256 * (k + 1)])
    assert np.isfinite(a).all()
    assert np.abs(np.linalg.norm(a[:, :, 3:], axis=2) - 1.0).max() < 1e-5
    # (4) A batch size is a scheduling decision, not a numerical one: 1280 frames take the other builds of k_solve (small + second launch instead of the single one) and of the cloud rows (one block per frame instead of two) -- and
    # must give the same poses bit for bit.
    B2 = 1280
    idx2 = np.arange(B2) % 256
    ctx = native.Context(ol.MODEL, B2)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        for kernel in (0, 2):      # the launcher's choice (the cooperative contact kernel), then the lane-per-pair kernel (the fall-back for models too large for the other's LDS)
            ctx.debug_contact_kernel(kernel)
            ctx.tracker_reset(d["startpose"][idx2])
            c = ctx.update_sync(d["depth"][idx2].reshape(B2, -1), d["cam"][idx2])
            assert ctx.capacity_events() == (0, 0, 0)
            for k in range(5):
                assert np.array_equal(a[:256], c[256 * k:256 * (k + 1)]), "contact kernel %d" % kernel
    finally:
        ctx.close()


def test_many_frames_through_the_reset_branch(weights):
    """How the full-reset branch is launched is a scheduling decision too.  With full_reset_on_error = 0 every frame of a 512-frame batch resets; the
    context counts them behind each update and, once an update had more reset frames than the device has CUs, launches the branch's many-frames
    organisation (two k_reset blocks per CU, four frames per contact block in those frames' first step) instead of the few-frames one.  The first
    update of a context can only take the few-frames one, the third has seen the counts: same inputs, same poses bit for bit."""
    from hand_tracking_samples_amd import native
    B = 512
    d = np.load(os.path.join(HERE, "golden", "frames256.npz"))
    idx = np.arange(B) % 256
    depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
    ctx = native.Context(ol.MODEL, B)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3, full_reset_on_error=0.0)
        out = []; org = []
        for _ in range(3):
            ctx.tracker_reset(start)
            out.append(ctx.update_sync(depth, cams))
            org.append(ctx.debug_reset_organisation())
        assert ctx.capacity_events() == (0, 0, 0)
        assert org[0] == 0 and org[2] == 1, org      # the many-frames organisation was really taken by the third update (its counts arrive asynchronously: the second may see them or not)
    finally:
        ctx.close()
    assert np.isfinite(out[0]).all()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
    assert np.array_equal(out[0][:256], out[0][256:])


def test_config4_shard_of_8192_frames(weights):
    """BASELINE configs[3]: one GPU's shard of the 65536-frame job = 8192 independent frames in one call (the 256 bench frames 32 times).
    (1) deterministic, (2) a frame's result does not depend on its slot (all 32 copies agree bit for bit), (3) no capacity of the kernels is
    touched, (4) a strided sample of 64 slots (64 distinct frames) agrees with the C restatement frame by frame."""
    from hand_tracking_samples_amd import native
    B = 8192
    d = np.load(os.path.join(HERE, "golden", "frames256.npz"))
    idx = np.arange(B) % 256
    depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
    ctx = native.Context(ol.MODEL, B)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(start)
        a = ctx.update_sync(depth, cams)
        ctx.tracker_reset(start)
        b = ctx.update_sync(depth, cams)
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()
    assert np.array_equal(a, b)
    for k in range(1, B // 256):
        assert np.array_equal(a[:256], a[256 * k:256 * (k + 1)]), "copy %d of the bench frames differs" % k
    assert np.isfinite(a).all() and np.abs(np.linalg.norm(a[:, :, 3:], axis=2) - 1.0).max() < 1e-5
    slots = np.arange(64) * 127                      # 64 slots spread over the shard; slot % 256 are 64 different frames
    assert len(set(slots % 256)) == 64 and slots.max() < B
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    ref = np.zeros((64, 17, 7), np.float32)
    for k, s in enumerate(slots):
        orc.reset(start[s])
        cam = ol.camera(cams[s])
        orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[s])), C.byref(cam), ol.fptr(ref[k]))
    orc.close()
    got = a[slots]
    dp = np.abs(got[:, :, :3] - ref[:, :, :3]).max(axis=(1, 2))
    dq = np.minimum(np.abs(got[:, :, 3:] - ref[:, :, 3:]), np.abs(got[:, :, 3:] + ref[:, :, 3:])).max(axis=(1, 2))
    tight = (dp <= 2e-5) & (dq <= 2e-4)
    loose = (dp <= 2e-4) & (dq <= 2e-3)
    print("8192-frame shard, 64 sampled slots: exact %d, tight %d, loose %d; worst |dpos| %.2e m |dquat| %.2e" % (int(((dp == 0) & (dq == 0)).sum()), int(tight.sum()), int(loose.sum()), dp.max(), dq.max()))
    assert loose.sum() >= 63 and tight.sum() >= 61
