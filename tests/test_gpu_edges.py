"""Edge cases of the per-frame path on the device against the C restatement (itself pinned bit for bit on the reference's goldens):
no point at all, a tile with every pixel in range (1024 points: the solver's single-body rows overflow their LDS pool into HBM),
every pixel of the tile as a point (sub-sampling off: 4096 points, the capacity), a 128x128 frame entirely in range (4096 points through
the full-frame path), a 320x240 frame entirely in range (19200 points, 76800 with sub-sampling off: the context's point capacity grows), and batch independence."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
POS_TOL, QUAT_TOL = 2e-4, 2e-3      # the CNN-accepted tolerance of tests/test_gpu_solver.py
TIGHT_POS_TOL, TIGHT_QUAT_TOL = 2e-5, 2e-4      # the solver tolerance of tests/test_gpu_solver.py (hand model on the CNN-independent branch)


def _bank(n=4):
    d = np.load(os.path.join(HERE, "golden", "frames256.npz"))
    idx = (np.arange(n) * 61) % len(d["depth"])
    return d["depth"][idx].reshape(n, 64, 64).copy(), d["cam"][idx].copy(), d["startpose"][idx].copy()


def _bumpy(h, w, seed, z0=0.42, amp=0.06):
    """a smooth bumpy surface entirely inside the depth range, in depth units of 1 mm"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    z = z0 + amp * np.sin(x / w * 5.0 + rng.uniform(0, 3)) * np.cos(y / h * 4.0 + rng.uniform(0, 3))
    return np.round(z * 1000.0).astype(np.uint16)


def _with_wall(depth, z_mm=650):
    """the frame's hand in front of a wall that is itself inside the depth range: every pixel becomes a point"""
    d = depth.copy()
    d[(d == 0) | (d > z_mm)] = z_mm
    return d


def _oracle_poses(weights, depth, cams, start, dims=(64, 64), fraction=4, model=None, nb=17, thr=0.0, passes=3):
    orc = ol.Oracle(weights, model=model)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = passes; orc.head.par.subsample_fraction = fraction; orc.head.par.accum_error_threshold = thr
    out = np.zeros((len(depth), nb, 7), np.float32); npts = []
    for k in range(len(depth)):
        orc.reset(start[k])
        cam = ol.camera(cams[k], dims[0], dims[1])
        orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[k])), C.byref(cam), ol.fptr(out[k]))
        npts.append(orc.flags()[2])
    orc.close()
    return out, npts


NEVER_ACCEPT = 1e9


def _compare(tag, got, ref, npts, tight=False):
    dp = np.abs(got[:, :, :3] - ref[:, :, :3]).max(axis=(1, 2))
    dq = np.minimum(np.abs(got[:, :, 3:] - ref[:, :, 3:]), np.abs(got[:, :, 3:] + ref[:, :, 3:])).max(axis=(1, 2))
    print("%s: points %s |dpos| %s |dquat| %s" % (tag, npts, ["%.1e" % v for v in dp], ["%.1e" % v for v in dq]))
    assert dp.max() <= (TIGHT_POS_TOL if tight else POS_TOL) and dq.max() <= (TIGHT_QUAT_TOL if tight else QUAT_TOL)


@pytest.fixture(scope="module")
def ctx(weights):
    from hand_tracking_samples_amd import native
    c = native.Context(ol.MODEL, 4)
    c.load_weights(weights)
    c.set_params(microforce=3.0, mainthreadpasses=3)
    yield c
    c.close()


def test_frame_without_points(ctx, weights):
    depth, cams, start = _bank(2)
    depth[0][:] = 0                              # nearer than 0.1 m: no pixel in range
    keep = np.zeros((64, 64), bool); keep[28:36, 28:36] = True
    depth[1][~keep] = 0                          # an 8x8 patch of the hand is all that is left: a handful of points
    ctx.tracker_reset(start)
    got = ctx.update_sync(depth, cams)
    ref, npts = _oracle_poses(weights, depth, cams, start)
    assert npts[0] == 0 and 0 < npts[1] <= 16
    _compare("no / few points", got, ref, npts)
    assert list(ctx.tracker_flags(2)[1]) == [50, 50]      # fewer than min_point_num points: initializing = 50 (handtrack.h:781-782)


def test_tile_entirely_in_range(ctx, weights):
    depth, cams, start = _bank(2)
    depth[0] = _bumpy(64, 64, 10); depth[1] = _with_wall(depth[1])
    # A wall of points makes the fit ill-conditioned: when the tracker takes over the CNN-driven pose, the 4e-4 the MFMA CNN's rounding
    # leaves on that pose grows to millimetres over the three passes (seen on frame 1).  NEVER_ACCEPT keeps the hand model on the
    # CNN-independent branch of handtrack.h:721, where only the solver's own rounding (Jacobian-form rows, csrc/ht_quad.hpp) separates the
    # device from the restatement.
    ctx.set_params(microforce=3.0, mainthreadpasses=3, accum_error_threshold=NEVER_ACCEPT)
    try:
        ctx.tracker_reset(start)
        got = ctx.update_sync(depth, cams)
    finally:
        ctx.set_params(microforce=3.0, mainthreadpasses=3, accum_error_threshold=0.0)
    ref, npts = _oracle_poses(weights, depth, cams, start, thr=NEVER_ACCEPT)
    assert npts == [1024, 1024]
    _compare("dense tile", got, ref, npts, tight=True)


def test_every_pixel_a_point(ctx, weights):
    depth, cams, start = _bank(2)
    depth[1] = _bumpy(64, 64, 20)      # frame 0: a hand with sub-sampling off, frame 1: 4096 points
    ctx.set_params(microforce=3.0, mainthreadpasses=3, subsample_fraction=1)
    try:
        ctx.tracker_reset(start)
        got = ctx.update_sync(depth, cams)
    finally:
        ctx.set_params(microforce=3.0, mainthreadpasses=3, subsample_fraction=4)
    ref, npts = _oracle_poses(weights, depth, cams, start, fraction=1)
    assert npts[1] == 4096
    _compare("fraction 1", got, ref, npts)


def test_full_frames_at_and_beyond_capacity(ctx, weights):
    from hand_tracking_samples_amd import native
    _, cams, start = _bank(1)
    cam128 = np.array([[163, 163, 64, 64, 0.001, 0, 0, 0, 0, 0, 0, 1]], np.float32)
    z = np.load(os.path.join(HERE, "golden", "frames5_64.npz"))      # 128x128 frames of the 26-bone hand; tracked here with the 17-bone model
    d128 = _with_wall(z["depth"][5])[None]
    start = z["startpose"][5:6, :17]
    # These all-in-range scenes are capacity tests, not hand data, and they are ill-conditioned: thousands of rows pull one body against a wall, and
    # each main-thread pass multiplies a rounding-level difference by 50-100 (restatement against the restatement with its rows evaluated in
    # Jacobian form, the device's arithmetic: 1e-7 m / 4e-6 after one pass, 7e-6 / 2e-4 after two, 7e-4 / 2e-2 after three on the 128x128 wall).
    # One pass sends every point through every kernel on the over-size code paths, which is what this test is about.
    ctx.set_params(microforce=3.0, mainthreadpasses=1, accum_error_threshold=NEVER_ACCEPT)      # see test_tile_entirely_in_range
    try:
        ctx.tracker_reset(start)
        got = ctx.update_frames_sync(d128, cam128, 0.17)
    finally:
        ctx.set_params(microforce=3.0, mainthreadpasses=3, accum_error_threshold=0.0)
    ref, npts = _oracle_poses(weights, d128, cam128, start, dims=(128, 128), thr=NEVER_ACCEPT, passes=1)
    assert npts == [4096] and ctx.frames_overflow() == 0
    _compare("128x128 in range", got, ref, npts, tight=True)
    # a 320x240 frame with every pixel in range: 19200 points after sub-sampling, 76800 without -- the context's point capacity grows, nothing is cut
    cam320 = np.array([[305, 305, 160, 120, 0.001, 0, 0, 0, 0, 0, 0, 1]], np.float32)
    d320 = _bumpy(240, 320, 31)[None]
    for fraction, want in ((4, 19200), (1, 76800)):
        ctx.set_params(microforce=3.0, mainthreadpasses=1, accum_error_threshold=NEVER_ACCEPT, subsample_fraction=fraction)
        try:
            ctx.tracker_reset(start)
            got = ctx.update_frames_sync(d320, cam320, 0.17)
        finally:
            ctx.set_params(microforce=3.0, mainthreadpasses=3, accum_error_threshold=0.0, subsample_fraction=4)
        ref, npts = _oracle_poses(weights, d320, cam320, start, dims=(320, 240), thr=NEVER_ACCEPT, fraction=fraction, passes=1)
        assert npts == [want] and ctx.frames_overflow() == 0 and ctx.point_capacity() == want
        _compare("320x240 in range, fraction %d" % fraction, got, ref, npts, tight=True)


def test_frames_of_a_batch_do_not_interact(ctx):
    depth, cams, start = _bank(4)
    ctx.tracker_reset(start)
    together = ctx.update_sync(depth, cams)
    for k in (0, 3):
        ctx.tracker_reset(start[k:k + 1])
        alone = ctx.update_sync(depth[k:k + 1], cams[k:k + 1])
        assert np.array_equal(alone[0], together[k])


def test_thumb_base_ignore_rewrite(weights, tmp_path):
    """HandModelEnhancements' one-time rewrite (handtrack.h:408-416): a model whose bone 2 ignores fewer than 10 bodies gets bone 2 out of every
    collision pair.  The stock hand never triggers it (12 distinct bodies), so the baked model is doctored: bone 2 keeps only its joint neighbours.
    Device and C restatement must agree on the contacts of a fist pose, and bone 2 must appear in none of them."""
    import htfx
    from hand_tracking_samples_amd import native
    sys_path = os.path.join(HERE, "..")
    import sys
    sys.path.insert(0, sys_path)
    from bench import _write_htfx
    m = htfx.load(ol.MODEL)
    ign = m["ignore"].copy()
    for j in range(17):
        if j not in (1, 3):
            ign[2, j] = 0; ign[j, 2] = 0
    assert ign[2].sum() == 2
    m2 = dict(m); m2["ignore"] = ign
    path = str(tmp_path / "hand_few_ignores.htfx")
    _write_htfx(path, m2)
    g = htfx.load(os.path.join(HERE, "golden", "golden8.htfx"))
    start = np.stack([g["f%d/startpose" % f] for f in range(8)])
    ctx = native.Context(path, 8)
    try:
        ctx.tracker_reset(start)
        c, n = ctx.stage_contacts(0, 8)
    finally:
        ctx.close()
    orc = ol.Oracle(weights, model=path)
    total = 0
    for f in range(8):
        orc.reset(start[f])
        buf = (ol.Contact * 96)()
        k = orc.L.ho_find_contacts(orc.h, orc.model(0), buf, 96)
        assert k == n[f], "frame %d: %d contacts on the device, %d in the restatement" % (f, n[f], k)
        for i in range(k):
            assert (int(c[f, i, 0]), int(c[f, i, 1])) == (buf[i].rb0, buf[i].rb1)
            assert 2 not in (buf[i].rb0, buf[i].rb1)
        total += k
    orc.close()
    assert total >= 20


def test_contact_patch_extra_samples(tmp_path):
    """ContactPatch's four extra samples (gjk.h:626-641: the shape tilted by 2 degrees about four axes through the first contact point, a sample kept
    when it lies 5 cm from every kept one).  The hand's bones are smaller than 5 cm, so no test on the hand reaches that code.  Here the box and the
    prism of the 3-body chain, scaled by 3 (16 x 11 x 27 cm and 7 x 16 cm) and with their joint's ignore entry cleared, are posed touching,
    penetrating and apart (body 2 leaves every pair through the rewrite of handtrack.h:408-416, as in any model the tracker loads).  Device contacts
    must equal the restatement's entry by entry, and some patch must really carry more than one sample."""
    import ctypes as C
    import sys
    import htfx
    from hand_tracking_samples_amd import native
    sys.path.insert(0, os.path.join(HERE, ".."))
    from bench import _write_htfx
    m = dict(htfx.load(os.path.join(HERE, "golden", "model_chain3.htfx")))
    ign = m["ignore"].copy(); ign[0, 1] = ign[1, 0] = 0; m["ignore"] = ign
    path = str(tmp_path / "chain3_collide.htfx")
    _write_htfx(path, m)
    ctx = native.Context(path, 16)
    orc = ol.Oracle(None, model=path)
    try:
        ctx.scale(3.0)
        orc.L.ho_scale.argtypes = [C.c_void_p, C.c_float]
        orc.L.ho_scale(orc.h, 3.0)
        rest = orc.get_state(0)
        def quat(axis, deg):
            a = np.asarray(axis, np.float64); a /= np.linalg.norm(a); h = np.deg2rad(deg) / 2
            return np.concatenate([a * np.sin(h), [np.cos(h)]]).astype(np.float32)
        poses = [((dx, 0, 0), quat((0, 0, 1), ang)) for ang, dx in ((-102.5, 0.1075), (-100.0, 0.11), (-97.5, 0.1125), (-95.0, 0.1125), (-92.5, 0.1125), (-90.0, 0.1175), (-90.0, 0.105), (20.0, 0.1125), (65.6, 0.1), (0.0, 0.13))]
        poses += [((0, 0.082, 0), quat((0, 0, 1), 30)), ((0.003, 0.088, -0.01), quat((0, 0, 1), 77.6)), ((0, 0, 0.2), quat((1, 0, 0), 1)), ((0.02, 0.01, 0.21), quat((0, 1, 0), 175))]
        states = []
        for pos, q in poses:
            s = rest.copy(); s[:, 7:] = 0
            s[2, :3] = (5.0, 5.0, 5.0)
            s[1, :3] = np.asarray(pos, np.float32) + rest[0, :3]; s[1, 3:7] = q
            states.append(s)
        states = np.stack(states)
        ctx.set_state(0, states)
        c, n = ctx.stage_contacts(0, len(states))
        most = total = 0
        for k in range(len(states)):
            orc.set_state(0, states[k])
            cs = (ol.Contact * 64)()
            mk = orc.L.ho_find_contacts(orc.h, orc.model(0), cs, 64)
            ref = np.array([[x.rb0, x.rb1, x.normal.x, x.normal.y, x.normal.z, x.p0w.x, x.p0w.y, x.p0w.z, x.p1w.x, x.p1w.y, x.p1w.z, x.separation] for x in cs[:mk]], np.float32).reshape(mk, 12)
            print("pose %d: %d contacts (separations %s)" % (k, mk, ["%.4f" % v for v in ref[:, 11]]))
            assert n[k] == mk
            assert np.array_equal(c[k, :mk], ref), "pose %d" % k
            most = max(most, mk); total += mk
        assert most >= 2 and total >= 14 and ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close(); orc.close()


def _clenched_model_and_state(tmp_path):
    """the stock hand with the ignore lists among palm and finger bones cleared (bone 2 keeps its own, so the rewrite of handtrack.h:408-416 stays off) and those sixteen
    bodies pushed into one another 5 cm above the palm's rest position: every pair of them touches"""
    import sys
    import htfx
    sys.path.insert(0, os.path.join(HERE, ".."))
    from bench import _write_htfx
    m = dict(htfx.load(ol.MODEL))
    ign = m["ignore"].copy()
    bodies = list(range(1, 17))
    for i in bodies:
        for j in bodies:
            if i != j and i != 2 and j != 2:
                ign[i, j] = 0
    m["ignore"] = ign
    path = str(tmp_path / "hand_clenched.htfx")
    _write_htfx(path, m)
    orc = ol.Oracle(None, model=path)
    rest = orc.get_state(0)
    s = rest.copy(); s[:, 7:] = 0
    rng = np.random.default_rng(7)
    for b in bodies:
        s[b, :3] = rest[1, :3] + np.array([0.0, 0.0, 0.05], np.float32) + rng.normal(0, 0.004, 3).astype(np.float32)
    return path, orc, s


def test_more_than_96_contacts_are_kept(tmp_path):
    """physics.h:451-462 keeps every contact FindShapeShapeContacts reports.  Until round 5 the device kept 96 per frame and counted the rest as dropped; now it keeps
    every touching sample its per-frame pool holds (192) and k_solve applies the groups beyond its level schedule's tables one at a time in row order.  A scene with 109
    contacts: (1) the contact list equals the restatement's entry by entry, from both contact kernels, and nothing is reported dropped; (2) FitPointCloud on that state
    (no points: joint rows and the 327 collision rows) agrees with the restatement to the solver's tolerance, as does the rest pose in the slot beside it.  The exact-order
    build through whole updates of this scene: tests/test_gpu_exact_solver.py."""
    from hand_tracking_samples_amd import native
    path, orc, s = _clenched_model_and_state(tmp_path)
    ctx = native.Context(path, 4)
    try:
        ordinary = orc.get_state(0).copy(); ordinary[:, 7:] = 0      # the rest pose
        orc.set_state(0, s)
        buf = (ol.Contact * 256)()
        k = orc.L.ho_find_contacts(orc.h, orc.model(0), buf, 256)
        assert 96 < k <= 192, k
        ref = np.array([[x.rb0, x.rb1, x.normal.x, x.normal.y, x.normal.z, x.p0w.x, x.p0w.y, x.p0w.z, x.p1w.x, x.p1w.y, x.p1w.z, x.separation] for x in buf[:k]], np.float32)
        states = np.stack([s, ordinary])
        for kernel in (0, 1, 2):      # the launcher's choice, cooperative, lane per pair
            ctx.debug_contact_kernel(kernel)
            ctx.set_state(0, states)
            c, n = ctx.stage_contacts(0, 2, cap=192)
            print("contact kernel %d: %d contacts on the device, %d in the restatement" % (kernel, n[0], k))
            assert n[0] == k
            assert np.array_equal(c[0, :k], ref)
            assert ctx.capacity_events() == (0, 0, 0)
        ctx.debug_contact_kernel(0)
        # FitPointCloud without points: joints + collisions
        empty = [np.zeros((0, 3), np.float32)] * 2
        none_l = [np.zeros((0, 16), np.float32)] * 2; none_a = [np.zeros((0, 8), np.float32)] * 2
        want = []
        for st in states:
            orc.set_state(0, st)
            A = (ol.Angular * 16)(); L = (ol.Linear * 1)(); nn = C.c_int(0); z = ol.F3(0, 0, 0)
            orc.L.ho_enhancements(orc.h, orc.model(0), A, C.byref(nn), 0, z, z, 0)
            orc.L.ho_fit_pointcloud(orc.h, orc.model(0), ol.f3ptr(np.zeros((1, 3), np.float32)), 0, L, 0, A, 0, 3.0)
            want.append(orc.get_state(0).copy())
        ctx.set_state(0, states)
        ctx.fit_rows(0, empty, none_l, none_a, microforce=3.0)
        got = ctx.get_state(0, 2)
        for f in range(2):
            dp = np.abs(got[f][:, :3] - want[f][:, :3]).max(); dq = np.abs(got[f][:, 3:7] - want[f][:, 3:7]).max(); dm = np.abs(got[f][:, 7:13] - want[f][:, 7:13]).max()
            print("FitPointCloud on %s: |dpos| %.2e |dquat| %.2e |dmom| %.2e" % ("109 contacts" if f == 0 else "the rest pose", dp, dq, dm))
            assert dp <= TIGHT_POS_TOL and dq <= TIGHT_QUAT_TOL
        assert ctx.capacity_events() == (0, 0, 0)
        ctx.debug_solver_build(0)
    finally:
        ctx.close(); orc.close()


def test_nan_in_the_tracked_state_ends_in_the_sanity_check_reset(ctx, weights):
    """SanityCheck (physmodel.h:437-442, called at the end of every FitPointCloud, :355): a body whose state holds a NaN is put back to its start pose with zero momenta.
    A NaN written into the hand model's carried state (one body's angular momentum; another tracker's wrist position) spreads through the joint rows to every body during
    the first pass's sweeps, the pass's SanityCheck resets them all, and the remaining passes track on from the model's rest pose.  Device against the restatement given
    the same poisoned state: the poses after one pass are the rest pose EXACTLY (nothing of the NaN survives, momenta zero), after three passes they agree to the
    solver's tolerance; the tracker flags agree too (FitError of a NaN pose is NaN, every comparison with it false: handtrack.h:704-722)."""
    depth, cams, start = _bank(4)
    poison = [(1, 5, 10), (2, 0, 0)]      # (tracker slot, body, component of [pos3 quat4 linmom3 angmom3])
    import htfx
    rest = htfx.load(ol.MODEL)["body_f"]      # per body: ... pos_start3 at [10:13], quat_start4 at [13:17]
    for passes in (1, 3):
        ctx.set_params(mainthreadpasses=passes)
        ctx.tracker_reset(start)
        st = ctx.get_state(0, 4)
        for slot, body, comp in poison:
            st[slot, body, comp] = np.nan
        ctx.set_state(0, st)
        got = ctx.update_sync(depth, cams)
        hand = ctx.get_state(0, 4)
        assert np.isfinite(got).all() and np.isfinite(hand).all()
        orc = ol.Oracle(weights)
        orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = passes
        ref = np.zeros((4, 17, 7), np.float32); ref_hand = np.zeros((4, 17, 13), np.float32); ref_flags = []
        for k in range(4):
            orc.reset(start[k]); s = orc.get_state(0)
            for slot, body, comp in poison:
                if slot == k:
                    s[body, comp] = np.nan
            orc.set_state(0, s)
            cam = ol.camera(cams[k])
            orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[k])), C.byref(cam), ol.fptr(ref[k]))
            ref_hand[k] = orc.get_state(0); ref_flags.append(orc.flags()[:2])
        orc.close()
        for slot, _, _ in poison:
            if passes == 1:      # every body back at its start pose, bit for bit, momenta zero: on the device and in the restatement
                for h in (hand[slot], ref_hand[slot]):
                    assert np.array_equal(h[:, :3], rest[:, 10:13]) and np.array_equal(h[:, 3:7], rest[:, 13:17]) and not h[:, 7:].any()
        _compare("NaN in the carried state, %d pass(es)" % passes, got, ref, [0] * 4)
        err, ini = ctx.tracker_flags(4)
        for k in range(4):
            assert ini[k] == ref_flags[k][1] and (err[k] == ref_flags[k][0] or (np.isnan(err[k]) and np.isnan(ref_flags[k][0]))), (k, err[k], ref_flags[k])
    ctx.set_params(mainthreadpasses=3)


def _chain_model_json(path, nb, full_ranges=True):
    """a chain of nb boxes along z in the reference's JSON schema (controlcages + joints), every joint with two-sided ranges on all three axes (six angular rows)"""
    import json
    rng = np.random.RandomState(7)
    cages, joints = [], []
    for b in range(nb):
        rx, ry, z1 = 0.012 + 0.0005 * (b % 3), 0.009, 0.02
        v = [(-rx, -ry, 0.0), (rx, -ry, 0.0), (rx, ry, 0.0), (-rx, ry, 0.0), (-rx, -ry, z1), (rx, -ry, z1), (rx, ry, z1), (-rx, ry, z1)]
        cages.append({"faces": [[3, 2, 1, 0], [4, 5, 6, 7], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7]],
                      "verts": [[float("%.6g" % (x + rng.uniform(-0.0005, 0.0005))) for x in p] for p in v]})
        if b:
            joints.append({"jointframe": [0, 0, 0, 1], "p0": [0, 0, 0.021], "p1": [0, 0, -0.001], "rangemax": [30, 20, 10] if full_ranges else [30, 0, 0],
                           "rangemin": [-30, -20, -10] if full_ranges else [-30, 0, 0], "rbi0": b - 1, "rbi1": b})
    with open(path, "w") as fp:
        json.dump({"controlcages": cages, "joints": joints}, fp)


def test_models_with_many_ranged_joints(weights, tmp_path):
    """The angular rows of a solve: 13 CNN-driven + up to 6 per joint.  A 27-body chain whose 26 joints are ranged on all three axes brings 13 + 156 = 169 rows
    (more than the 126 the ordinary solver builds keep): ht_launch_solve takes the build with four row slots per lane (252 rows, records beyond 126 in HBM), no
    capacity event.  With the exact-order sweeps the update equals the restatement BIT FOR BIT (same rows, same order, nothing dropped); the product's sweeps agree
    to the solver's tolerance after one MultiStepSim step and one pass on every frame that the restatement itself does not carry a 1e-7 nudge of the heat-maps
    beyond that tolerance on (a thin 27-link chain thrown at a hand's cloud amplifies rounding quickly: 2e-3 after two steps).
    A model with 27 joints is refused by ht_create (13 + 9 per joint would not fit 252)."""
    from hand_tracking_samples_amd import native
    ok_json, big_json, baked = str(tmp_path / "chain27.json"), str(tmp_path / "chain28.json"), str(tmp_path / "chain27.htfx")
    _chain_model_json(ok_json, 27); _chain_model_json(big_json, 28)
    with pytest.raises(native.HTError, match="26 joints"):
        native.Context(big_json, 1)
    native.model_bake(ok_json, baked)
    depth, cams, _ = _bank(2)
    start = np.zeros((2, 27, 7), np.float32); start[:, :, 6] = 1.0
    for b in range(27):      # the chain laid out in front of the camera, inside the cloud
        start[:, b, :3] = (-0.05 + 0.004 * b, 0.01 * np.sin(0.4 * b), 0.45 + 0.002 * b)
    L = ol.lib()
    for build in (0, 5):
        ctx = native.Context(ok_json, 2)
        try:
            assert (ctx.nb, ctx.nj) == (27, 26)
            ctx.load_weights(weights)
            ctx.set_params(microforce=3.0, mainthreadpasses=1, steps=1)
            if build:
                ctx.debug_solver_build(build)
            ctx.tracker_reset(start)
            got, cnn = ctx.update_sync(depth, cams, want_cnn=True)
            other = ctx.get_state(1, 2)
            an = ctx.cnn_results(2)[2]
            assert ctx.capacity_events() == (0, 0, 0)
        finally:
            ctx.close()
        orc = ol.Oracle(weights, model=baked)
        orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 1; orc.head.par.steps = 1
        ref = np.zeros((2, 27, 7), np.float32); ref_other = np.zeros((2, 27, 13), np.float32)
        L.ho_set_round_once(1)
        try:
            for k in range(2):
                y = np.ascontiguousarray(cnn[k]); L.ho_set_cnn_override(orc.h, ol.fptr(y))      # the same heat-maps on both sides: this test is about the solver
                orc.reset(start[k]); cam = ol.camera(cams[k])
                L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[k])), C.byref(cam), ol.fptr(ref[k]))
                ref_other[k] = orc.get_state(1)
        finally:
            L.ho_set_cnn_override(orc.h, None); L.ho_set_round_once(0); orc.close()
        if build == 5:
            assert np.array_equal(got, ref) and np.array_equal(other, ref_other), "exact-order sweeps on 169 angular rows"
        else:
            # The product's sweeps are another floating-point build of the same rows (csrc/ht_quad.hpp), and this scene carries a rounding-sized change to the end of
            # the update at full size: the RESTATEMENT run on heat-maps nudged by 1e-7 of their peak ends 7e-3 m / 0.5 (quaternion) away from itself on either frame
            # (a hand's heat-maps driving a 27-link chain with force 10000, after a full reset).  So the whole update is held to the restatement with the exact-order
            # sweeps above, and the product's sweeps over the 169 rows where a tolerance means something: one MultiStepSim step from the same state, below.
            assert np.isfinite(other).all() and np.isfinite(got).all() and np.abs(np.linalg.norm(got[:, :, 3:], axis=2) - 1.0).max() < 1e-5
            analysis = an
    # one MultiStepSim step (cloud rows, contacts, ONE solve over 13 + 156 angular rows) from the same state with the product's sweeps and with the exact-order ones
    one = {}
    for build in (0, 5):
        ctx = native.Context(ok_json, 2)
        try:
            ctx.load_weights(weights)
            ctx.set_params(microforce=3.0, mainthreadpasses=1, steps=1)
            if build:
                ctx.debug_solver_build(build)
            ctx.stage_prepare(depth, cams); ctx.tracker_reset(start)
            ctx.stage_multistep(analysis, 2)
            one[build] = ctx.get_state(1, 2)
            assert ctx.capacity_events() == (0, 0, 0)
        finally:
            ctx.close()
    dp = np.abs(one[0][:, :, :3] - one[5][:, :, :3]).max(); dq = np.abs(one[0][:, :, 3:7] - one[5][:, :, 3:7]).max()
    print("27-body chain, one step over 169 angular rows: the product's sweeps against the exact-order ones |dpos| %.1e m |dquat| %.1e" % (dp, dq))
    assert dp <= TIGHT_POS_TOL and dq <= TIGHT_QUAT_TOL


def test_contact_pool_cannot_overflow_for_the_stock_hand():
    """physics.h:451-462 keeps every contact in a std::vector; the contact kernel keeps the touching samples of a frame in a pool of 192 (40 five-sample patches).  For the stock
    17-bone hand that is provably enough: 91 pairs of bodies collide without ignoring each other, and none of them joins two bodies both large enough for ContactPatch's four extra
    samples (gjk.h:625-641 against the 0.05 m proximity test) -- at most 91 samples, no patch.  configs[4]'s 26-bone hand has 244 such pairs: there the pool is a counted capacity."""
    from hand_tracking_samples_amd import native
    ctx = native.Context(ol.MODEL, 1)
    try:
        samples, patches, pool, slots = ctx.contact_capacity()
        assert (samples, patches) == (91, 0) and samples <= pool == 192 and patches <= slots
    finally:
        ctx.close()
    ctx = native.Context(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_hand26.htfx"), 1)
    try:
        samples, patches, pool, slots = ctx.contact_capacity()
        assert samples == 244 and patches == 0 and samples > pool
    finally:
        ctx.close()
