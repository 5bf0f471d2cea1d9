"""The solver is the one stage of the device path that is deliberately NOT the reference's operation order (rows in Jacobian form with fused multiply-adds,
DESIGN.md section 4).  This file isolates it: `ht_debug_solver_build(ctx, 5)` swaps the sweeps of k_solve -- and of the single-body solves inside k_reset --
for the reference's own Iter functions in the reference's row order (physics.h:251-265, 289-307, 556-581), one lane per frame, no fused multiply-adds; every
other kernel, every row builder, the level-independent state handling and the launch sequence stay the product's.  With that instantiation an update must
equal the CPU restatement -- which reproduces the reference bit for bit (tests/test_oracle_vs_golden.py) -- BIT FOR BIT on every frame, once both are given
the same CNN output (the net accumulates on MFMA tiles; its own parity is tests/test_gpu_cnn.py).  What then remains between the product build and the
reference is the Jacobian-form arithmetic of the sweeps alone, and the frames it moves out of the tight band are named with the amplification that moves them."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FR = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))      # the bench's 1024 distinct frames
REF = htfx.load(os.path.join(HERE, "golden", "poses1024.htfx"))
N = len(FR["depth"])


@pytest.fixture(scope="module")
def weights():
    return W.make_cnnb()


def _restatement_with_cnn(weights, cnn_out, updates=1, model=None, depth=None, cams=None, start=None, wh=(64, 64), direct=None):
    """the CPU restatement's unit of work per frame, fed with the given CNN outputs: user poses, othermodel states, flags"""
    depth = FR["depth"] if depth is None else depth; cams = FR["cam"] if cams is None else cams; start = FR["startpose"] if start is None else start
    orc = ol.Oracle(weights if direct is None else None, model=model)
    if direct is not None:
        assert orc.L.ho_set_direct(orc.h, direct[0], ol.fptr(direct[1]), direct[1].size) == 0
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    orc.L.ho_set_round_once(1)      # float sin / cos / acos rounded once from double, as the device forms them (glibc's float functions differ in the last bit now and then)
    n = len(depth)
    user = np.zeros((updates, n, orc.nb, 7), np.float32); other = np.zeros((updates, n, orc.nb, 13), np.float32); hand = np.zeros((updates, n, orc.nb, 13), np.float32); flags = np.zeros((updates, n, 2), np.float32)
    try:
        for i in range(n):
            orc.reset(start[i])
            cam = ol.camera(cams[i], wh[0], wh[1])
            for u in range(updates):
                y = np.ascontiguousarray(cnn_out[u][i]); orc.L.ho_set_cnn_override(orc.h, ol.fptr(y))
                orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[i]).reshape(-1)), C.byref(cam), ol.fptr(user[u, i]))
                other[u, i] = orc.get_state(1); hand[u, i] = orc.get_state(0); flags[u, i] = orc.flags()[:2]
        orc.L.ho_set_cnn_override(orc.h, None)
    finally:
        orc.L.ho_set_round_once(0)
        orc.close()
    return user, other, hand, flags


def test_exact_order_solver_reproduces_the_restatement_bit_for_bit_on_all_1024_frames(weights):
    from hand_tracking_samples_amd import native
    ctx = native.Context(ol.MODEL, N)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.debug_solver_build(5)
        ctx.tracker_reset(FR["startpose"])
        cnn = []; poses = []; others = []; hands = []; flags = []; resets = []
        for u in range(2):      # two consecutive updates: the second carries momenta, prev_frame_error and `initializing`
            p, c = ctx.update_sync(FR["depth"].reshape(N, -1), FR["cam"], want_cnn=True)
            poses.append(p); cnn.append(c); others.append(ctx.get_state(1, N)); hands.append(ctx.get_state(0, N)); flags.append(np.stack(ctx.tracker_flags(N), 1).astype(np.float32))
            resets.append(ctx.debug_reset_flags(N))
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()
    user, other, hand, fl = _restatement_with_cnn(weights, cnn, updates=2)
    for u in range(2):
        bad_other = [i for i in range(N) if not np.array_equal(others[u][i], other[u, i])]
        bad_hand = [i for i in range(N) if not np.array_equal(hands[u][i], hand[u, i])]
        bad_user = [i for i in range(N) if not np.array_equal(poses[u][i], user[u, i])]
        print("update %d, exact-order solver against the restatement given the device's CNN output: %d / %d / %d of %d frames differ (othermodel / handmodel / user poses)" % (u, len(bad_other), len(bad_hand), len(bad_user), N))
        assert not bad_other and not bad_hand and not bad_user, (u, bad_other[:8], bad_hand[:8], bad_user[:8])
        assert np.array_equal(flags[u], fl[u])
    # the full-reset branch (PoseFromScratch + UnibodyFit's single-body solves, handtrack.h:706-711) was among them: the device's own decision flags
    print("frames through the full-reset branch: first update %s, second update %s" % (np.nonzero(resets[0])[0].tolist(), np.nonzero(resets[1])[0].tolist()))
    assert resets[0].sum() >= 4


def test_exact_order_solver_under_the_products_launch_sequence(weights):
    """ht_debug_solver_build 5 takes an update's kernels in order on one stream; 8 keeps the PRODUCT's choreography (the carried pose's FitError and the reset decision
    beside the net, the reset frames' chain lapped with the batch's first two steps, rows phases forked over three streams) around the same exact-order sweeps: the two must
    agree bit for bit on every frame, i.e. the stream choreography changes no result."""
    from hand_tracking_samples_amd import native
    ctx = native.Context(ol.MODEL, N)
    res = {}
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        for build in (5, 8):
            ctx.debug_solver_build(build)
            ctx.tracker_reset(FR["startpose"])
            out = []
            for u in range(2):
                p = ctx.update_sync(FR["depth"].reshape(N, -1), FR["cam"])
                out.append((p, ctx.get_state(1, N), ctx.get_state(0, N), np.stack(ctx.tracker_flags(N), 1), ctx.debug_reset_flags(N)))
            res[build] = out
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.debug_solver_build(0)
        ctx.close()
    for u in range(2):
        for a, b in zip(res[5][u], res[8][u]):
            assert np.array_equal(a, b), "update %d" % u
    assert res[8][0][4].sum() >= 4      # reset frames took the lapped chain


def test_exact_order_solver_with_more_than_96_contacts(weights, tmp_path):
    """physics.h:451-462 keeps every contact.  The clenched scene of tests/test_gpu_edges.py (109 contacts at the start) through two whole updates -- the job's five
    MultiStepSim steps and the three main passes all solve more contacts than k_solve's level schedule has tables for: with the exact-order sweeps the device equals the
    restatement bit for bit, nothing is dropped."""
    from hand_tracking_samples_amd import native
    from test_gpu_edges import _clenched_model_and_state
    path, orc, s = _clenched_model_and_state(tmp_path)
    orc.close()
    depth = np.stack([FR["depth"][0].reshape(-1), np.zeros(4096, np.uint16)])
    cams = np.stack([FR["cam"][0], FR["cam"][0]]); start = np.stack([s[:, :7], s[:, :7]])
    ctx = native.Context(path, 2)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.debug_solver_build(5)
        ctx.tracker_reset(start)
        c0, n0 = ctx.stage_contacts(0, 2, cap=192)
        assert n0.min() > 96
        cnn = []; poses = []; others = []; hands = []
        for u in range(2):
            p, c = ctx.update_sync(depth, cams, want_cnn=True)
            poses.append(p); cnn.append(c); others.append(ctx.get_state(1, 2)); hands.append(ctx.get_state(0, 2))
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()
    user, other, hand, _ = _restatement_with_cnn(weights, cnn, updates=2, model=path, depth=depth, cams=cams, start=start)
    for u in range(2):
        for i in range(2):
            assert np.array_equal(others[u][i], other[u, i]) and np.array_equal(hands[u][i], hand[u, i]) and np.array_equal(poses[u][i], user[u, i]), (u, i)
    print("more than 96 contacts (%s at the start), exact-order solver against the restatement: two updates bit for bit" % n0.tolist())


def test_exact_order_solver_on_config5_end_to_end_26_bones():
    """The same statement for BASELINE configs[4] end to end (tests/test_config5_e2e.py): 128x128 frames, the 128x128-input net, 26 bones -- 325 body pairs, two
    chain rounds per sweep, the cloned fingers in permanent contact (15 expanding-polytope runs per frame).  Exact-order sweeps: bit for bit on all 64 frames."""
    from hand_tracking_samples_amd import native
    fr = np.load(os.path.join(HERE, "golden", "frames5_64.npz"))
    model26 = os.path.join(HERE, "golden", "model_hand26.htfx")
    w128 = W.make_cnnb128()
    n = len(fr["depth"])
    ctx = native.Context(model26, n)
    try:
        ctx.load_weights128(w128)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.debug_solver_build(5)
        ctx.tracker_reset(fr["startpose"])
        poses, cnn = ctx.update_direct_sync(fr["depth"], fr["cam"], 128, want_cnn=True)
        others = ctx.get_state(1, n); hands = ctx.get_state(0, n)
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()
    user, other, hand, _ = _restatement_with_cnn(None, [cnn], updates=1, model=model26, depth=fr["depth"], cams=fr["cam"], start=fr["startpose"], wh=(128, 128), direct=(128, w128))
    bad = [i for i in range(n) if not (np.array_equal(others[i], other[0, i]) and np.array_equal(hands[i], hand[0, i]) and np.array_equal(poses[i], user[0, i]))]
    print("config 5 end to end, exact-order solver against the restatement given the device's CNN output: %d of %d frames differ" % (len(bad), n))
    assert not bad, bad


def test_exact_order_solver_follows_a_stream_bit_for_bit(weights):
    """Deep tracker states.  Sixty-four trackers each follow a moving hand for twelve consecutive updates -- tracker i sees bench frame (16 i + k) mod 1024 in update k, i.e.
    every update brings a NEW frame nine animation rows on (the other tests show a tracker the same frame twice) -- so the carried pose lags the cloud, the accumulated-error
    take (handtrack.h:712-726), the `initializing` countdown (:727) and the full-reset branch (:706-711) fire in the middle of a stream, on states no fixture starts from.
    Free-running, no teacher forcing: with the exact-order sweeps the device equals the restatement -- given the device's heat-maps -- bit for bit after EVERY update: user
    poses, both models' states with momenta, prev_frame_error and `initializing`."""
    from hand_tracking_samples_amd import native
    T, K = 64, 12
    idx = [(16 * np.arange(T) + k) % N for k in range(K)]
    depth = FR["depth"].reshape(N, -1); start = FR["startpose"][idx[0]]
    ctx = native.Context(ol.MODEL, T)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.debug_solver_build(5)
        ctx.tracker_reset(start)
        dev = []
        for k in range(K):
            p, c = ctx.update_sync(depth[idx[k]], FR["cam"][idx[k]], want_cnn=True)
            dev.append((p, c, ctx.get_state(1, T), ctx.get_state(0, T), np.stack(ctx.tracker_flags(T), 1).astype(np.float32), ctx.debug_reset_flags(T)))
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.debug_solver_build(0)
        ctx.close()
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    orc.L.ho_set_round_once(1)
    bad = []; takes = 0
    try:
        for i in range(T):
            orc.reset(start[i])
            for k in range(K):
                f = idx[k][i]
                cam = ol.camera(FR["cam"][f], 64, 64)
                orc.L.ho_set_cnn_override(orc.h, ol.fptr(np.ascontiguousarray(dev[k][1][i])))
                user = np.zeros((orc.nb, 7), np.float32)
                before = orc.get_state(0)
                orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[f])), C.byref(cam), ol.fptr(user))
                same = (np.array_equal(dev[k][0][i], user) and np.array_equal(dev[k][2][i], orc.get_state(1)) and np.array_equal(dev[k][3][i], orc.get_state(0))
                        and np.array_equal(dev[k][4][i], np.array(orc.flags()[:2], np.float32)))
                if not same:
                    bad.append((i, k)); break      # a tracker that left the restatement's stream stays out: report where it left
        orc.L.ho_set_cnn_override(orc.h, None)
    finally:
        orc.L.ho_set_round_once(0)
        orc.close()
    resets = np.stack([d[5] for d in dev])      # [K, T]
    ini = np.stack([d[4][:, 1] for d in dev])
    print("stream of %d updates on %d trackers, exact-order solver against the restatement: %d trackers leave it (first at %s); full resets per update %s; trackers with `initializing` > 0 after the last update: %d"
          % (K, T, len(bad), bad[:4], resets.sum(axis=1).tolist(), int((ini[-1] > 0).sum())))
    assert not bad, bad
    assert resets[1:].sum() >= 8      # the full-reset branch fired in mid-stream, on several trackers


def test_exact_order_solver_follows_a_stream_on_config5():
    """The stream test on BASELINE configs[4] end to end (128x128 frames, the 128x128-input net, 26 bones): sixteen trackers, six updates, tracker i sees frame (16 i + k) mod 256
    of bench_data/frames5_256.npz in update k.  Exact-order sweeps, free-running: bit for bit after every update."""
    from hand_tracking_samples_amd import native
    fr = np.load(os.path.join(ROOT, "bench_data", "frames5_256.npz"))
    model26 = os.path.join(HERE, "golden", "model_hand26.htfx")
    w128 = W.make_cnnb128()
    n5 = len(fr["depth"]); T, K = 16, 6
    idx = [(16 * np.arange(T) + k) % n5 for k in range(K)]
    start = fr["startpose"][idx[0]]
    ctx = native.Context(model26, T)
    try:
        ctx.load_weights128(w128)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.debug_solver_build(5)
        ctx.tracker_reset(start)
        dev = []
        for k in range(K):
            p, c = ctx.update_direct_sync(fr["depth"][idx[k]], fr["cam"][idx[k]], 128, want_cnn=True)
            dev.append((p, c, ctx.get_state(1, T), ctx.get_state(0, T), np.stack(ctx.tracker_flags(T), 1).astype(np.float32), ctx.debug_reset_flags(T)))
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.debug_solver_build(0)
        ctx.close()
    orc = ol.Oracle(None, model=model26)
    assert orc.L.ho_set_direct(orc.h, 128, ol.fptr(w128), w128.size) == 0
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    orc.L.ho_set_round_once(1)
    bad = []
    try:
        for i in range(T):
            orc.reset(start[i])
            for k in range(K):
                f = idx[k][i]
                cam = ol.camera(fr["cam"][f], 128, 128)
                orc.L.ho_set_cnn_override(orc.h, ol.fptr(np.ascontiguousarray(dev[k][1][i])))
                user = np.zeros((orc.nb, 7), np.float32)
                orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(fr["depth"][f]).reshape(-1)), C.byref(cam), ol.fptr(user))
                same = (np.array_equal(dev[k][0][i], user) and np.array_equal(dev[k][2][i], orc.get_state(1)) and np.array_equal(dev[k][3][i], orc.get_state(0))
                        and np.array_equal(dev[k][4][i], np.array(orc.flags()[:2], np.float32)))
                if not same:
                    bad.append((i, k)); break
        orc.L.ho_set_cnn_override(orc.h, None)
    finally:
        orc.L.ho_set_round_once(0)
        orc.close()
    resets = np.stack([d[5] for d in dev])
    print("configs[4] stream of %d updates on %d trackers, exact-order solver against the restatement: %d trackers leave it (first at %s); full resets per update %s"
          % (K, T, len(bad), bad[:4], resets.sum(axis=1).tolist()))
    assert not bad, bad


# HandTracker's public knobs (handtrack.h:560-603; config file keys handtrack.h:846-880) and PhysModel's (physmodel.h:229-236), each moved off its default.
# device name -> (oracle struct: "par" | "phys", oracle field)
_KNOB = {"physics_iterations": ("phys", "iterations"), "physics_iterations_post": ("phys", "iterations_post"), "physics_use_collision": ("phys", "use_collision"),
         "physics_weak_force": ("phys", "weak_force"), "bone_sum_error_scale": ("phys", "bone_sum_error_scale"), "unibody_force": ("phys", "unibody_force")}
_VARIANTS = [
    ("defaults of the class (microforce 1, one pass)", dict(microforce=1.0, mainthreadpasses=1)),
    ("angles_only", dict(angles_only=1)),
    ("no boundary planes", dict(boundary_planes=0)),
    ("always_take_cnn", dict(always_take_cnn=1)),
    ("four steps, cloud from step 0", dict(steps=4, steps_keypoints=2, steps_keyangles=1, steps_palmangle=1, steps_cloudstart=0)),
    ("seven steps, cloud from step 3, one unibody step", dict(steps=7, steps_keypoints=5, steps_keyangles=3, steps_palmangle=1, steps_cloudstart=3, steps_unibody=1)),
    ("every second point, few points needed", dict(subsample_fraction=2, min_point_num=100)),
    ("many points needed (initializing stays armed)", dict(min_point_num=2000)),
    ("accumulated-error threshold, ray probability floor", dict(accum_error_threshold=0.05, min_cray_prob=0.1)),
    ("weak cloud force", dict(cloudforce_max_point=5.0, cloudforce_max_sum=500.0)),
    ("eager full reset", dict(full_reset_on_error=0.2)),
    ("reluctant full reset", dict(full_reset_on_error=5.0)),
    ("short depth range", dict(drangey=0.55)),
    ("eight iterations, two post", dict(physics_iterations=8, physics_iterations_post=2)),
    ("no collisions", dict(physics_use_collision=0)),
    ("joint-limit force, error scale, unibody force", dict(physics_weak_force=0.2, bone_sum_error_scale=2.0, unibody_force=0.3)),
    ("voxel subsampling", dict(subsample_voxel=1, subsample_size=0.01, subsample_fraction=1)),
    ("five passes, strong microforce", dict(mainthreadpasses=5, microforce=6.0)),
]


@pytest.mark.parametrize("name,knobs", _VARIANTS, ids=[v[0].split(",")[0].replace(" ", "_") for v in _VARIANTS])
def test_exact_order_solver_follows_a_stream_with_every_knob_moved(weights, name, knobs):
    """Every public parameter of the tracker moved off its default, one group at a time, through a four-update stream on 32 trackers: exact-order sweeps, free-running,
    against the restatement with the same parameter -- bit for bit after every update.  (The sweeps are the only thing the product build does differently, and a parameter
    reaches them as data: what this pins is every parameter's way through the host side and the row-building kernels, on states a stream produces.  The product's sweeps
    under the same parameter are held against the exact-order ones on the stream's first update.)"""
    from hand_tracking_samples_amd import native
    T, K = 32, 4
    idx = [(32 * np.arange(T) + 5 + k) % N for k in range(K)]
    depth = FR["depth"].reshape(N, -1); start = FR["startpose"][idx[0]]
    par = dict(microforce=3.0, mainthreadpasses=3); par.update(knobs)
    ctx = native.Context(ol.MODEL, T)
    try:
        ctx.load_weights(weights)
        ctx.set_params(**par)
        ctx.debug_solver_build(5)
        ctx.tracker_reset(start)
        dev = []
        for k in range(K):
            p, c = ctx.update_sync(depth[idx[k]], FR["cam"][idx[k]], want_cnn=True)
            dev.append((p, c, ctx.get_state(1, T), ctx.get_state(0, T), np.stack(ctx.tracker_flags(T), 1).astype(np.float32), ctx.debug_reset_flags(T)))
        assert ctx.capacity_events() == (0, 0, 0)
        # the PRODUCT sweeps under the same parameter: the stream's first update again (one update: little room for amplification).  A parameter the product's sweeps
        # dropped or misread would move every tracker; rounding moves the median by nothing and single trackers by an amplified last bit.
        ctx.debug_solver_build(0)
        ctx.tracker_reset(start)
        prod = ctx.update_sync(depth[idx[0]], FR["cam"][idx[0]])
        import parity_rule as pr
        dp, dq = pr.pose_diff(prod, dev[0][0])
        tight = int(((dp <= pr.TIGHT[0]) & (dq <= pr.TIGHT[1])).sum())
        print("%-55s: product sweeps against exact-order sweeps, first update: median %.1e m / %.1e, %d of %d trackers inside 2e-5 m / 2e-4, largest %.1e m / %.1e" % (name, np.median(dp), np.median(dq), tight, T, dp.max(), dq.max()))
        assert np.isfinite(prod).all() and np.median(dp) <= 2e-6 and np.median(dq) <= 4e-5 and tight >= T - 4 and dp.max() <= pr.CAP_TAKE_CNN[0] and dq.max() <= pr.CAP_TAKE_CNN[1]
    finally:
        ctx.debug_solver_build(0)
        ctx.close()
    orc = ol.Oracle(weights)
    for k_, v in par.items():
        where, field = _KNOB.get(k_, ("par", k_))
        setattr(orc.head.par if where == "par" else orc.head.phys, field, v)
    orc.L.ho_set_round_once(1)
    bad = []
    try:
        for i in range(T):
            orc.reset(start[i])
            for k in range(K):
                f = idx[k][i]
                cam = ol.camera(FR["cam"][f], 64, 64)
                orc.L.ho_set_cnn_override(orc.h, ol.fptr(np.ascontiguousarray(dev[k][1][i])))
                user = np.zeros((orc.nb, 7), np.float32)
                orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[f])), C.byref(cam), ol.fptr(user))
                same = (np.array_equal(dev[k][0][i], user) and np.array_equal(dev[k][2][i], orc.get_state(1)) and np.array_equal(dev[k][3][i], orc.get_state(0))
                        and np.array_equal(dev[k][4][i], np.array(orc.flags()[:2], np.float32)))
                if not same:
                    bad.append((i, k)); break
        orc.L.ho_set_cnn_override(orc.h, None)
    finally:
        orc.L.ho_set_round_once(0)
        orc.close()
    resets = np.stack([d[5] for d in dev])
    print("%-55s: %d of %d trackers leave the restatement's stream (first at %s); full resets per update %s" % (name, len(bad), T, bad[:4], resets.sum(axis=1).tolist()))
    assert not bad, bad
