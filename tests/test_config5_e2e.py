"""BASELINE configs[4] END TO END as SURVEY 8(d) "config 5 (i)-(iii)" defines it: a 128x128 frame of the 26-bone hand goes through the
128x128-input net, the heat-map decode with camsub(cam, 8) and the tracker (FitError, reset branch, MultiStepSim, accept, three FitPointCloud
passes) in ONE unit of work -- the stages of update_cnn_model_threadsafe (handtrack.h:693-729) called directly on the frame, HandSegmentVR
bypassed (a frame of the net's own size is its own segment, :283-284).  tests/golden/e2e128.htfx comes from the reference's own functions
and layer classes (`HT_REF_MODEL_JSON=... ref_harness e2e128`, tools/regen_goldens.sh): per-stage dumps of four frames, a second update on
the carried state, one start far off with always_take_cnn (full-reset branch + accepted CNN pose), and the results for all 64 bench frames."""
import ctypes as C
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL26 = os.path.join(HERE, "golden", "model_hand26.htfx")
G = htfx.load(os.path.join(HERE, "golden", "e2e128.htfx"))
FR = np.load(os.path.join(HERE, "golden", "frames5_64.npz"))
IDX = [int(i) for i in G["frames"]]
POS_TOL, QUAT_TOL, FULL_POS_TOL, FULL_QUAT_TOL = 2e-5, 2e-4, 2e-4, 2e-3      # as tests/test_gpu_solver.py states them


@pytest.fixture(scope="module")
def weights128():
    assert float(np.float32(W.DEFAULT_SEED)) == G["weights_seed_gain"][0] and G["weights_seed_gain"][1] == W.DEFAULT_FC2_GAIN
    return W.make_cnnb128()


@pytest.fixture(scope="module")
def oracle(weights128):
    orc = ol.Oracle(None, model=MODEL26)
    assert orc.L.ho_set_direct(orc.h, 128, ol.fptr(weights128), weights128.size) == 0
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    yield orc
    orc.close()


def _update(orc, i):
    user = np.zeros((26, 7), np.float32)
    cam = ol.camera(FR["cam"][i], 128, 128)
    orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(FR["depth"][i])), C.byref(cam), ol.fptr(user))
    return user


@pytest.mark.parametrize("k", range(len(IDX)))
def test_oracle_runs_config5_end_to_end_like_the_reference(oracle, k):
    pre = "f%d/" % k
    oracle.reset(FR["startpose"][IDX[k]])
    user = _update(oracle, IDX[k])
    assert np.array_equal(oracle.get_state(1), G[pre + "uw_other_final"])
    assert np.array_equal(oracle.get_state(0), G[pre + "uw_hand_pass2"])
    assert np.array_equal(user, G[pre + "uw_pose_user"])
    if k == 0:      # the second update carries momenta and tracker flags
        user2 = _update(oracle, IDX[k])
        assert np.array_equal(oracle.get_state(0), G[pre + "second/uw_hand_pass2"])
        assert np.array_equal(user2, G[pre + "second/uw_pose_user"])
    if k + 1 == len(IDX):      # far-off start, always_take_cnn: the reset branch and the accept
        assert G[pre + "far/errors"][0] > 0.6 and G[pre + "far/uw_accept"][0] == 26
        oracle.reset(G[pre + "far/startpose"]); oracle.head.par.always_take_cnn = 1
        try:
            user3 = _update(oracle, IDX[k])
        finally:
            oracle.head.par.always_take_cnn = 0
        assert np.array_equal(oracle.get_state(1), G[pre + "far/uw_other_after_cnn"])
        assert np.array_equal(user3, G[pre + "far/uw_pose_user"])


def test_oracle_reproduces_all_64_bench_frames(oracle):
    for i in range(len(FR["depth"])):
        oracle.reset(FR["startpose"][i])
        user = _update(oracle, i)
        assert np.array_equal(user, G["all/uw_pose_user"][i]), i
        assert np.array_equal(oracle.get_state(1)[:, :7], G["all/other_pose"][i]), i
        assert oracle.flags()[:2] == (G["all/flags"][i][0], int(G["all/flags"][i][1])), i


G256 = htfx.load(os.path.join(HERE, "golden", "e2e128_256.htfx"))      # the bench's configs[4] batch: 256 distinct frames, the reference's results for all of them (all/ only)
FR256 = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames5_256.npz"))


def test_oracle_reproduces_all_256_bench_frames(oracle):
    """what bench.py --workload config5-e2e verifies against: the restatement reproduces the reference on every one of the 256 frames, bit for bit"""
    user = np.zeros((26, 7), np.float32)
    for i in range(len(FR256["depth"])):
        oracle.reset(FR256["startpose"][i])
        cam = ol.camera(FR256["cam"][i], 128, 128)
        oracle.L.ho_update(oracle.h, ol.u16ptr(np.ascontiguousarray(FR256["depth"][i])), C.byref(cam), ol.fptr(user))
        assert np.array_equal(user, G256["all/uw_pose_user"][i]), i
        assert np.array_equal(oracle.get_state(1)[:, :7], G256["all/other_pose"][i]), i
        assert oracle.flags()[:2] == (G256["all/flags"][i][0], int(G256["all/flags"][i][1])), i


@pytest.mark.gpu
def test_gpu_config5_end_to_end_on_the_256_bench_frames(weights128):
    """the same unit of work on the bench's 256 distinct frames against the reference's results: flags identical, user poses in the bands, othermodel's median at rounding level"""
    from hand_tracking_samples_amd import native
    n = len(FR256["depth"])
    ctx = native.Context(MODEL26, n)
    try:
        ctx.load_weights128(weights128)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(FR256["startpose"])
        poses, _ = ctx.update_direct_sync(FR256["depth"], FR256["cam"], 128, want_cnn=True)
        other = ctx.get_state(1, n)[:, :, :7]
        ref = G256["all/uw_pose_user"]
        dp = np.abs(poses[:, :, :3] - ref[:, :, :3]).max(axis=(1, 2)); dq = np.abs(poses[:, :, 3:] - ref[:, :, 3:]).max(axis=(1, 2))
        do = np.abs(other - G256["all/other_pose"]).max(axis=(1, 2))
        err, init = ctx.tracker_flags(n)
        out = np.nonzero((dp > POS_TOL) | (dq > QUAT_TOL))[0]
        print("config 5 end to end, 256 frames: user poses |dpos| max %.2e median %.2e, |dquat| max %.2e, %d outside the tight band %s; othermodel max %.2e p90 %.2e median %.2e"
              % (dp.max(), np.median(dp), dq.max(), len(out), out.tolist(), do.max(), np.percentile(do, 90), np.median(do)))
        assert np.isfinite(poses).all() and np.isfinite(other).all()
        assert np.array_equal(init, G256["all/flags"][:, 1].astype(np.int32))
        assert np.abs(err - G256["all/flags"][:, 0]).max() <= 1e-4
        # Against how far the reference's OWN FMA builds move these frames (tests/golden/ref_spread5e2e_256.npz).  This model -- cloned fingers in permanent contact, 15 polytope
        # runs per frame -- amplifies a rounding difference on about a tenth of its frames in every build, and WHICH frames depends on the perturbation (the reference's two FMA
        # builds disagree with each other there), so the frames outside the band are held by number and size (tests/parity_rule.py's bands and cap), not by name; that nothing
        # but rounding is at work is shown bit for bit by the exact-order build (tests/test_gpu_exact_solver.py).
        import parity_rule as pr
        spread = np.load(os.path.join(HERE, "golden", "ref_spread5e2e_256.npz"))
        for key, dev, rf in (("user", poses, ref), ("other", other, G256["all/other_pose"])):
            ok, d = pr.distribution(dev, rf, *pr.spread_of(spread, key), cap=pr.CAP_TAKE_CNN, max_factor=2.0)
            print("  %s: %s" % (key, d))
            assert ok, key
        assert np.median(do) <= FULL_POS_TOL
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_config5_end_to_end_matches_reference(weights128):
    """Device: k_cnn_input<128> -> k_conv1<128> -> k_conv2 -> k_fc(K = 12544) -> k_fc144 -> decode (camsub 8) -> FitError / reset / MultiStepSim / accept
    -> three main passes on the 26-bone model, through ht_update_direct_sync; all 64 bench frames against the reference's results."""
    from hand_tracking_samples_amd import native
    n = len(FR["depth"])
    ctx = native.Context(MODEL26, n)
    try:
        ctx.load_weights128(weights128)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(FR["startpose"])
        poses, cnn = ctx.update_direct_sync(FR["depth"], FR["cam"], 128, want_cnn=True)
        for k, i in enumerate(IDX):
            assert np.abs(cnn[i] - G["f%d/cnn_output" % k]).max() <= 2e-5
        other = ctx.get_state(1, n)[:, :, :7]
        ref = G["all/uw_pose_user"]
        dp = np.abs(poses[:, :, :3] - ref[:, :, :3]).max(axis=(1, 2)); dq = np.abs(poses[:, :, 3:] - ref[:, :, 3:]).max(axis=(1, 2))
        do = np.abs(other - G["all/other_pose"]).max(axis=(1, 2))
        err, init = ctx.tracker_flags(n)
        print("config 5 end to end, 64 frames: user poses |dpos| max %.2e median %.2e, |dquat| max %.2e; othermodel (CNN-driven) max %.2e median %.2e" % (dp.max(), np.median(dp), dq.max(), do.max(), np.median(do)))
        assert np.array_equal(init, G["all/flags"][:, 1].astype(np.int32))
        assert np.abs(err - G["all/flags"][:, 0]).max() <= 1e-4
        # no frame of the set accepts the CNN pose, so the hand model never sees the MFMA-rounded net: the tight band, except where the three ill-conditioned
        # passes amplify the solver's rounding (Jacobian-form rows) -- those frames are listed; how many and how far is held against the reference's own builds on the 256-frame set
        out = np.nonzero((dp > POS_TOL) | (dq > QUAT_TOL))[0]
        print("  outside the tight band: %s" % ", ".join("frame %d |dpos| %.2e |dquat| %.2e" % (i, dp[i], dq[i]) for i in out))
        # how many and how far, derived from the fixture (the 64 frames are frames 0, 4, 8, ... of the 256-frame set, tests/golden/ref_spread5e2e_256.npz): frame by frame
        # by tests/parity_rule.py's rule -- outside the tight band only where the reference's own FMA builds are, by at most twice their move -- except on as many frames as
        # those builds fail each other's rule on (parity_rule.cross_build_failures); nothing beyond twice those builds' largest move on the 256 frames
        import parity_rule as pr
        spread = np.load(os.path.join(HERE, "golden", "ref_spread5e2e_256.npz"))
        sub = np.arange(n) * 4
        assert np.array_equal(FR["depth"], FR256["depth"][sub])
        sp, sq = pr.spread_of(spread, "user", sub)
        okf, _ = pr.frame_rule(dp, dq, sp, sq, 2.0, pr.CAP_TAKE_CNN)
        allowed = pr.cross_build_failures(spread, "user", 2.0, pr.CAP_TAKE_CNN, sub)
        print("  frames failing the per-frame rule: %s; the reference's own FMA builds held against each other on these 64 frames: %d" % (np.nonzero(~okf)[0].tolist(), allowed))
        spa, sqa = pr.spread_of(spread, "user")      # how far a failing frame may go: twice the largest move of the reference's own builds on this model's frames (parity_rule.distribution's bound)
        assert int((~okf).sum()) <= allowed and dp.max() <= 2 * float(spa.max()) and dq.max() <= 2 * float(sqa.max()) and np.median(dp) <= 1e-6
        # othermodel is driven hard by the decoded angles of the MFMA-accumulated net (MultiStepSim, 10000 N drives) on a model whose cloned fingers sit in permanent
        # contact with the originals (15 polytope runs per frame): half the frames stay at rounding level, the others amplify it -- in the reference's own FMA builds
        # just as much.  That nothing but rounding separates the two is shown bit for bit by tests/test_gpu_exact_solver.py (the same 64 frames, exact-order sweeps).
        assert np.median(do) <= FULL_POS_TOL
        assert ctx.capacity_events() == (0, 0, 0)
        # a second update on the carried state, and the far-off start with always_take_cnn (reset branch + accepted CNN pose) on the last listed frame
        k = len(IDX) - 1; i = IDX[k]
        far = np.repeat(G["f%d/far/startpose" % k][None], n, 0)
        ctx.set_params(always_take_cnn=1)
        ctx.tracker_reset(far)
        poses3, _ = ctx.update_direct_sync(np.repeat(FR["depth"][i][None], n, 0), np.repeat(FR["cam"][i][None], n, 0), 128)
        ref3 = G["f%d/far/uw_pose_user" % k]
        dp3 = np.abs(poses3[:, :, :3] - ref3[None, :, :3]).max(); dq3 = np.abs(poses3[:, :, 3:] - ref3[None, :, 3:]).max()
        print("far-off start, CNN pose accepted after the full reset: |dpos| %.2e |dquat| %.2e" % (dp3, dq3))
        assert np.array_equal(poses3[0], poses3[n - 1])      # slot-independent
        assert dp3 <= FULL_POS_TOL and dq3 <= FULL_QUAT_TOL
    finally:
        ctx.close()
