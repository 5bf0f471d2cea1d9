"""Pins the CPU oracle (oracle/*.c) to the reference: every stage is compared with tests/golden/golden8.htfx,
which oracle/_ref/ref_harness produced by running the reference's own code on the same inputs.

Both sides are built IEEE / -ffp-contract=off, the oracle keeps the reference's evaluation order and calls the same
glibc, so the comparison is bit for bit (np.array_equal) unless stated otherwise.  CPU only.
"""
import ctypes as C

import numpy as np
import pytest

import htfx
import oracle_lib as ol

NFRAMES = 8


@pytest.fixture(scope="module")
def orc(weights):
    o = ol.Oracle(weights)
    o.head.par.microforce = 3.0        # synthetic-tracker.cpp:91-93
    o.head.par.mainthreadpasses = 3
    yield o
    o.close()


def _frame(golden, f):
    pre = "f%d/" % f
    cam = ol.camera(golden[pre + "cam"])
    depth = np.ascontiguousarray(golden[pre + "depth"].reshape(-1))
    return pre, cam, depth


def _state_from_pose(pose7):
    s = np.zeros((17, 13), np.float32)
    s[:, :7] = pose7
    return s


def _vpts(orc, depth, cam):
    pts = np.zeros((4096, 3), np.float32)
    nfull = C.c_int(0)
    n = orc.L.ho_pointcloud(ol.u16ptr(depth), C.byref(cam), 0.1, 0.7, 4, ol.f3ptr(pts), 4096, C.byref(nfull))
    return np.ascontiguousarray(pts[:n]), nfull.value


def _analysis(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    hcam = ol.camera(golden[pre + "cam"], 16, 16)
    hcam.focal.x /= 4.0; hcam.focal.y /= 4.0; hcam.principal.x /= 4.0; hcam.principal.y /= 4.0
    an = ol.Analysis()
    out = np.ascontiguousarray(golden[pre + "cnn_output"])
    orc.L.ho_decode(ol.fptr(out), C.byref(hcam), C.byref(an))
    return an


def test_cnn_layers_bit_exact(orc, golden, weights):
    pre, cam, depth = _frame(golden, 0)
    inp = np.zeros(4096, np.float32)
    orc.L.ho_cnn_input(ol.u16ptr(depth), 4096, cam.depth_scale, 0.1, 0.7, ol.fptr(inp))
    assert np.array_equal(inp, golden[pre + "cnn_input"])
    sizes = (57600, 57600, 14400, 3600, 9216, 9216, 2304, 2048, 2048, 2304, 2304)
    layers = [np.zeros(n, np.float32) for n in sizes]
    arr = (C.POINTER(C.c_float) * 11)(*[ol.fptr(a) for a in layers])
    out = np.zeros(2304, np.float32)
    orc.L.ho_cnn_eval(ol.fptr(weights), ol.fptr(inp), ol.fptr(out), arr)
    for li in (3, 6, 7, 8, 9, 10):
        assert np.array_equal(layers[li], golden[pre + "cnn_layer%d" % li]), "layer %d" % li
    assert np.array_equal(out, golden[pre + "cnn_output"])
    assert abs(out.sum() - 24.0) < 1e-3          # 24 softmax chunks


@pytest.mark.parametrize("f", range(NFRAMES))
def test_cnn_output_bit_exact(orc, golden, weights, f):
    pre, cam, depth = _frame(golden, f)
    inp = np.zeros(4096, np.float32)
    orc.L.ho_cnn_input(ol.u16ptr(depth), 4096, cam.depth_scale, 0.1, 0.7, ol.fptr(inp))
    out = np.zeros(2304, np.float32)
    orc.L.ho_cnn_eval(ol.fptr(weights), ol.fptr(inp), ol.fptr(out), None)
    assert np.array_equal(out, golden[pre + "cnn_output"])


@pytest.mark.parametrize("f", range(NFRAMES))
def test_decode(orc, golden, f):
    pre = "f%d/" % f
    an = _analysis(orc, golden, f)
    crays = np.array([[c.x, c.y, c.z, c.w] for c in an.crays], np.float32)
    ip = np.array([[p.x, p.y] for p in an.image_points], np.float32)
    assert np.array_equal(crays, golden[pre + "an_crays"])
    assert np.array_equal(ip, golden[pre + "an_image_points"])
    assert np.array_equal(np.array(an.confidence[:], np.float32), golden[pre + "an_confidence"])
    assert np.array_equal(np.array(an.vals[:], np.float32), golden[pre + "an_vals"])
    ang = np.array([an.wristroll, an.pitch, an.tilt, an.palmq.x, an.palmq.y, an.palmq.z, an.palmq.w], np.float32)
    assert np.array_equal(ang, golden[pre + "an_angles"])
    assert np.array_equal(np.array(an.finger_clenched[:], np.float32), golden[pre + "an_clenched"])


@pytest.mark.parametrize("f", range(NFRAMES))
def test_pointcloud_and_fiterror(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    vpts, nfull = _vpts(orc, depth, cam)
    assert [nfull, len(vpts)] == golden[pre + "pc_count"].tolist()
    assert np.array_equal(vpts, golden[pre + "vpts"])
    orc.set_state(0, _state_from_pose(golden[pre + "startpose"]))
    err = orc.L.ho_fit_error(orc.h, orc.model(0), ol.f3ptr(vpts), len(vpts), ol.u16ptr(depth), C.byref(cam))
    assert np.float32(err) == golden[pre + "fiterror_start"][0]


@pytest.mark.parametrize("f", range(NFRAMES))
def test_closest_and_cloud_rows(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    vpts = golden[pre + "vpts"]
    orc.set_state(0, _state_from_pose(golden[pre + "startpose"]))
    m = orc.model(0)
    got = np.zeros((len(vpts), 5), np.float32)
    for i, v in enumerate(vpts):
        pl = ol.F4()
        got[i, 0] = orc.L.ho_closest(m, ol.v3(v), C.byref(pl))
        got[i, 1:] = [pl.x, pl.y, pl.z, pl.w]
    assert np.array_equal(got, golden[pre + "closest_vpts"])
    origin = ol.v3(golden[pre + "cam"][5:8])
    rows = (ol.Linear * len(vpts))()
    sub = vpts[::4]
    for i, v in enumerate(sub):
        rows[i] = orc.L.ho_cloud_constraint(m, ol.v3(v), origin)
    assert np.array_equal(ol.linears_to_array(rows, len(sub)), golden[pre + "cloud_rows_sub"])
    if (pre + "cloud_rows_vpts") in golden:
        for i, v in enumerate(vpts):
            rows[i] = orc.L.ho_cloud_constraint(m, ol.v3(v), origin)
        assert np.array_equal(ol.linears_to_array(rows, len(vpts)), golden[pre + "cloud_rows_vpts"])


@pytest.mark.parametrize("f", range(NFRAMES))
def test_joint_rows_enhancements_and_chamber(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    orc.set_state(0, _state_from_pose(golden[pre + "startpose"]))
    m = orc.model(0)
    ang = (ol.Angular * 256)()
    n = C.c_int(0)
    z = ol.F3(0, 0, 0)
    orc.L.ho_enhancements(orc.h, m, ang, C.byref(n), 0, z, z, 0)
    assert n.value == 0
    lin = (ol.Linear * 256)()
    nl = orc.L.ho_joint_linears(m, lin)
    assert np.array_equal(ol.linears_to_array(lin, nl), golden[pre + "joint_linears"])
    na = orc.L.ho_joint_angulars(orc.h, m, ang)
    assert np.array_equal(ol.angulars_to_array(ang, na), golden[pre + "joint_angulars"])
    # HandModelEnhancements with the palm / arm directions MultiStepSim passes (camera pose identity here)
    n = C.c_int(0)
    orc.L.ho_enhancements(orc.h, m, ang, C.byref(n), 0, ol.F3(-1, 0, 0), ol.F3(0, -1, 0), 0)
    assert np.array_equal(ol.angulars_to_array(ang, n.value), golden[pre + "enh_angulars"])
    an = _analysis(orc, golden, f)
    na = orc.L.ho_apply_angles(orc.h, m, C.byref(an), cam.pose, 10000.0, 10.0, ang)
    assert np.array_equal(ol.angulars_to_array(ang, na), golden[pre + "apply_angles"])
    vpts = np.ascontiguousarray(golden[pre + "vpts"])
    nl = orc.L.ho_cloud_chamber(m, ol.f3ptr(vpts), len(vpts), lin, 10.0)
    assert np.array_equal(ol.linears_to_array(lin, nl), golden[pre + "chamber_rows"])


@pytest.mark.parametrize("f", range(NFRAMES))
def test_contacts(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    orc.set_state(0, _state_from_pose(golden[pre + "startpose"]))
    cs = (ol.Contact * 256)()
    n = orc.L.ho_find_contacts(orc.h, orc.model(0), cs, 256)
    got = np.zeros((n, 18), np.float32)
    for i in range(n):
        c = cs[i]
        got[i] = [c.rb0, c.rb1, c.normal.x, c.normal.y, c.normal.z, c.p0w.x, c.p0w.y, c.p0w.z, c.p1w.x, c.p1w.y, c.p1w.z, c.separation,
                  c.p0.x, c.p0.y, c.p0.z, c.p1.x, c.p1.y, c.p1.z]
    assert np.array_equal(got, golden[pre + "contacts_start"])


def test_gjk_epa_cases(orc, golden):
    """48 hand-made body pairs: far, near/touching and penetrating (EPA path, hull.h:233-310)."""
    cin, cout = golden["gjk_cases_in"], golden["gjk_cases_out"]
    m = orc.model(0)
    s = orc.get_state(0)
    npen = 0
    for i in range(len(cin)):
        a, b = int(cin[i, 0]), int(cin[i, 1])
        s[a, :7] = cin[i, 2:9]
        s[b, :7] = cin[i, 9:16]
        orc.set_state(0, s)
        pa, pb = orc.L.ho_body_ptr(m, a), orc.L.ho_body_ptr(m, b)
        h = orc.L.ho_separated_bodies(pa, pb)
        got = np.array([h.normal.x, h.normal.y, h.normal.z, h.p0w.x, h.p0w.y, h.p0w.z, h.p1w.x, h.p1w.y, h.p1w.z, h.separation], np.float32)
        assert np.array_equal(got, cout[i, :10]), "case %d" % i
        hits = (ol.GjkContact * 5)()
        cnt = orc.L.ho_contact_patch_bodies(pa, pb, np.float32(0.03 / 8.0), hits)
        assert cnt == int(cout[i, 10])
        for k in range(cnt):
            gk = np.array([hits[k].p0w.x, hits[k].p0w.y, hits[k].p0w.z, hits[k].p1w.x, hits[k].p1w.y, hits[k].p1w.z, hits[k].separation], np.float32)
            assert np.array_equal(gk, cout[i, 11 + 7 * k: 18 + 7 * k])
        npen += h.separation <= 0
    assert npen >= 8      # the penetrating cases really exercised EPA


@pytest.mark.parametrize("f", range(NFRAMES))
def test_fit_pointcloud_two_passes(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    vpts = np.ascontiguousarray(golden[pre + "vpts"])
    orc.reset(golden[pre + "startpose"])
    m = orc.model(0)
    z = ol.F3(0, 0, 0)
    for p in range(2):
        ang = (ol.Angular * 16)()
        n = C.c_int(0)
        orc.L.ho_enhancements(orc.h, m, ang, C.byref(n), 0, z, z, 0)
        orc.L.ho_fit_pointcloud(orc.h, m, ol.f3ptr(vpts), len(vpts), None, 0, ang, n.value, 3.0)
        assert np.array_equal(orc.get_state(0), golden[pre + "fit_pass%d" % p]), "pass %d" % p


@pytest.mark.parametrize("f", range(NFRAMES))
def test_multistep(orc, golden, f):
    pre, cam, depth = _frame(golden, f)
    vpts = np.ascontiguousarray(golden[pre + "vpts"])
    an = _analysis(orc, golden, f)
    for s in range(1, 6):
        orc.reset(golden[pre + "startpose"])
        orc.head.par.steps = s
        orc.L.ho_multistep(orc.h, orc.model(1), C.byref(an), ol.f3ptr(vpts), len(vpts), cam.pose)
        assert np.array_equal(orc.get_state(1), golden[pre + "multistep%d" % s]), "steps=%d" % s
    orc.head.par.steps = 5


@pytest.mark.parametrize("f", range(3))
def test_reset_path(orc, golden, f):
    """PoseFromScratch + 3x UnibodyFit (handtrack.h:706-711)."""
    pre, cam, depth = _frame(golden, f)
    vpts = np.ascontiguousarray(golden[pre + "vpts"])
    an = _analysis(orc, golden, f)
    orc.reset(golden[pre + "startpose"])
    m = orc.model(1)
    orc.L.ho_pose_from_scratch(orc.h, m, ol.f3ptr(vpts), len(vpts), C.byref(an), cam.pose)
    assert np.array_equal(orc.get_state(1), golden[pre + "scratch"])
    for i in range(3):
        orc.L.ho_unibody_fit(orc.h, m, ol.f3ptr(vpts), len(vpts), cam.pose.position)
        assert np.array_equal(orc.get_state(1), golden[pre + "unibody%d" % i]), "unibody %d" % i
    err = orc.L.ho_fit_error(orc.h, m, ol.f3ptr(vpts), len(vpts), ol.u16ptr(depth), C.byref(cam))
    assert np.float32(err) == golden[pre + "fiterror_scratch"][0]


@pytest.mark.parametrize("f", range(NFRAMES))
def test_unit_of_work_two_frames(orc, golden, f):
    """update_cnn_model + 3 passes, then the same image again with carried state (handtrack.h:693-785)."""
    pre, cam, depth = _frame(golden, f)
    orc.reset(golden[pre + "startpose"])
    user = np.zeros((17, 7), np.float32)
    orc.L.ho_update(orc.h, ol.u16ptr(depth), C.byref(cam), ol.fptr(user))
    assert np.array_equal(user, golden[pre + "uw_pose_user"])
    assert np.array_equal(orc.get_state(0), golden[pre + "uw_hand_pass2"])
    t = ol.TrackerHead.from_address(orc.h)
    orc.L.ho_update(orc.h, ol.u16ptr(depth), C.byref(cam), ol.fptr(user))
    assert np.array_equal(user, golden[pre + "uw2_pose_user"])
    assert np.array_equal(orc.get_state(0), golden[pre + "uw2_hand"])


def test_unit_of_work_on_all_256_bench_frames(weights):
    """The C restatement against the reference on every frame the bench and the batch parity tests use (tests/golden/poses256.htfx = the reference's
    user poses, othermodel poses and tracker flags after the whole unit of work, `ref_harness poses`): bit for bit."""
    import os
    d = np.load(os.path.join(ol.GOLDEN, "frames256.npz"))
    ref = htfx.load(os.path.join(ol.GOLDEN, "poses256.htfx"))
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    user = np.zeros((17, 7), np.float32)
    for k in range(256):
        orc.reset(d["startpose"][k])
        cam = ol.camera(d["cam"][k])
        orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(d["depth"][k].reshape(-1))), C.byref(cam), ol.fptr(user))
        assert np.array_equal(user, ref["uw_pose_user"][k]), "frame %d: user pose" % k
        assert np.array_equal(orc.get_state(1)[:, :7], ref["other_pose"][k]), "frame %d: othermodel" % k
        e, i, n = orc.flags()
        assert (np.float32(e), i, n) == (ref["flags"][k, 0], int(ref["flags"][k, 1]), int(ref["flags"][k, 2])), "frame %d: flags" % k
    orc.close()


def test_unit_of_work_on_all_1024_bench_frames(weights):
    """The same on the 1024 DISTINCT frames bench.py times (bench_data/frames1024.npz, poses1024.htfx = `ref_harness posesfull`): the checker the device is held
    to on every frame of the headline number reproduces the reference on every one of them, bit for bit."""
    import os
    d = np.load(os.path.join(ol.ROOT, "bench_data", "frames1024.npz"))
    ref = htfx.load(os.path.join(ol.GOLDEN, "poses1024.htfx"))
    n = len(d["depth"])
    assert n == 1024 and ref["uw_pose_user"].shape[0] == n
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    user = np.zeros((17, 7), np.float32)
    bad = []
    for k in range(n):
        orc.reset(d["startpose"][k])
        cam = ol.camera(d["cam"][k])
        orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(d["depth"][k].reshape(-1))), C.byref(cam), ol.fptr(user))
        e, i, _ = orc.flags()
        if not (np.array_equal(user, ref["uw_pose_user"][k]) and np.array_equal(orc.get_state(1)[:, :7], ref["other_pose"][k]) and (np.float32(e), i) == (ref["flags"][k, 0], int(ref["flags"][k, 1]))):
            bad.append(k)
    orc.close()
    assert not bad, "frames on which the restatement differs from the reference: %s" % bad[:16]


def test_unit_of_work_always_take_cnn(weights):
    """The application's always_take_cnn switch (synthetic-tracker.cpp:91; handtrack.h:720-722): the restatement against `ref_harness poses ... takecnn` on every fourth of
    the 1024 bench frames (tests/golden/poses1024_takecnn.htfx): user poses, othermodel and flags bit for bit.  Every frame takes the accept branch here."""
    import os
    d = np.load(os.path.join(ol.ROOT, "bench_data", "frames1024.npz"))
    ref = htfx.load(os.path.join(ol.GOLDEN, "poses1024_takecnn.htfx"))
    plain = htfx.load(os.path.join(ol.GOLDEN, "poses1024.htfx"))
    assert not np.array_equal(ref["uw_pose_user"], plain["uw_pose_user"])      # the switch matters
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3; orc.head.par.always_take_cnn = 1
    user = np.zeros((17, 7), np.float32)
    for k in range(0, 1024, 4):
        orc.reset(d["startpose"][k])
        cam = ol.camera(d["cam"][k])
        orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(d["depth"][k].reshape(-1))), C.byref(cam), ol.fptr(user))
        assert np.array_equal(user, ref["uw_pose_user"][k]), "frame %d: user pose" % k
        assert np.array_equal(orc.get_state(1)[:, :7], ref["other_pose"][k]), "frame %d: othermodel" % k
        e, i, n = orc.flags()
        assert (np.float32(e), i, n) == (ref["flags"][k, 0], int(ref["flags"][k, 1]), int(ref["flags"][k, 2])), "frame %d: flags" % k
    orc.close()


def test_the_rounding_mode_of_the_exact_order_comparison_moves_frames_at_rounding_level_only(weights):
    """tests/test_gpu_exact_solver.py runs the restatement with ho_set_round_once(1): its three float libm calls (sinf / cosf / acosf in quat_axis_angle, ConstrainAngularDrive,
    ConstrainConeAngle) rounded once from double, as the device forms them -- NOT the pinned mode (glibc's float functions, as the reference calls them).  How many frames the
    two modes differ on, and by how much: every eighth of the 1024 bench frames."""
    import os
    d = np.load(os.path.join(ol.ROOT, "bench_data", "frames1024.npz"))
    ref = htfx.load(os.path.join(ol.GOLDEN, "poses1024.htfx"))
    orc = ol.Oracle(weights)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    user = np.zeros((17, 7), np.float32)
    idx = list(range(0, 1024, 8)); differ = 0; dmax = 0.0; dmed = []
    try:
        orc.L.ho_set_round_once(1)
        for k in idx:
            orc.reset(d["startpose"][k])
            cam = ol.camera(d["cam"][k])
            orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(d["depth"][k].reshape(-1))), C.byref(cam), ol.fptr(user))
            dd = float(np.abs(user - ref["uw_pose_user"][k]).max())
            differ += dd != 0.0; dmax = max(dmax, dd); dmed.append(dd)
    finally:
        orc.L.ho_set_round_once(0)
        orc.close()
    print("round-once libm against the pinned mode: %d of %d frames differ, median |d| %.1e, max %.1e" % (differ, len(idx), float(np.median(dmed)), dmax))
    assert np.median(dmed) <= 2e-6 and dmax <= 5e-3
