"""The reference-named C++ surface (include/ht_handtrack.hpp, global names) driven the way synthetic-hand-tracker/synthetic-tracker.cpp drives the
reference (tests/cxx_headless_driver.cpp follows its call sequence :90-96,139,204-216 without the window), compared with what the reference
itself produced:

  fakedepth   LoadHandModel + PhysModel::SetPose + HitCheck per pixel (the application's software rasteriser, :69-76) on the host
              vs the 320x240 depth frames the reference rendered (tests/golden/fullframe320.htfx)
  track       HandSegmentVR, camsub, GatherHandExpectedCNN, HandTracker::update, cnn_output, handmodel.GetPoseUser on the device
              vs the reference's poses for the same frames (tests/golden/golden8.htfx `uw_pose_user`)"""
import os
import struct
import subprocess

import numpy as np
import pytest

import htfx
import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
POS_TOL, QUAT_TOL = 2e-4, 2e-3      # whole path with the CNN on MFMA, as tests/test_gpu_solver.py


def _build(tmp_path):
    from hand_tracking_samples_amd import native
    native.load()
    exe = str(tmp_path / "driver")
    lib = os.path.dirname(native.lib_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "cxx_headless_driver.cpp"), "-o", exe, "-L" + lib, "-lht_mi355x", "-Wl,-rpath," + lib])
    return exe


def _write_input(path, depth, cams, start, gt):
    n, h, w = depth.shape
    nb = start.shape[1]
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", n, w, h, nb))
        for k in range(n):
            f.write(np.ascontiguousarray(depth[k], np.uint16).tobytes()); f.write(np.ascontiguousarray(cams[k], np.float32).tobytes())
            f.write(np.ascontiguousarray(start[k], np.float32).tobytes()); f.write(np.ascontiguousarray(gt[k], np.float32).tobytes())


def test_fake_hand_depth_raster_matches_the_reference(tmp_path):
    """host only: the model facade (ht_model_open / ht_model_hitcheck behind PhysModel::HitCheck) renders the frames the reference rendered"""
    exe = _build(tmp_path)
    G = htfx.load(os.path.join(HERE, "golden", "fullframe320.htfx"))
    n = len(G["rows"])
    depth = np.stack([G["f%d/depth" % f] for f in range(n)]); cams = np.stack([G["f%d/cam" % f] for f in range(n)])
    gt = np.stack([G["f%d/gtpose" % f] for f in range(n)]); start = np.stack([G["f%d/startpose" % f] for f in range(n)])
    _write_input(tmp_path / "in.bin", depth, cams, start, gt)
    out = subprocess.check_output([exe, "fakedepth", ol.MODEL, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")]).decode()
    assert "512 mesh triangles" in out      # GetMeshes(true): the subdivision surface of the palm (258 vertices, 256 quads)
    got = np.fromfile(tmp_path / "out.bin", np.uint16).reshape(depth.shape)
    hand = depth < 4000
    assert hand.sum() > 2000
    assert np.array_equal(got, depth), "%d of %d pixels differ" % ((got != depth).sum(), depth.size)


def test_point_cloud_free_function_matches_the_reference(tmp_path, golden):
    """host only: PointCloud(dimage, range) (misc_image.h:409-417, drawn at synthetic-tracker.cpp:233); every 4th point of it is the tracker's cloud
    (physmodel.h:58-64), which the reference dumped per golden frame (`vpts`)."""
    exe = _build(tmp_path)
    nf = 8
    depth = np.stack([golden["f%d/depth" % f] for f in range(nf)]); cams = np.stack([golden["f%d/cam" % f] for f in range(nf)])
    start = np.stack([golden["f%d/startpose" % f] for f in range(nf)])
    _write_input(tmp_path / "in.bin", depth, cams, start, start)
    out = subprocess.check_output([exe, "pointcloud", "-", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")]).decode()
    assert "pointcloud: 8 frames" in out
    raw = open(tmp_path / "out.bin", "rb").read()
    off = 0
    for f in range(nf):
        n = struct.unpack_from("<i", raw, off)[0]; off += 4
        pts = np.frombuffer(raw, np.float32, 3 * n, off).reshape(n, 3); off += 12 * n
        assert n == int(golden["f%d/pc_count" % f][0])
        assert np.array_equal(pts[::4], golden["f%d/vpts" % f]), "frame %d" % f


def test_the_applications_include_block_compiles_against_the_compat_headers(tmp_path):
    """include/compat/ staged where the reference's include/ and third_party/ stand: the application's own include lines (synthetic-tracker.cpp:15-24, tracker side) and its own
    LoadAnimBank compile and link unchanged, and the animation bank it reads through the binding's Pose extraction has the reference's 2336 rows when the bank is present."""
    import shutil
    from hand_tracking_samples_amd import native
    native.load()
    lib = os.path.dirname(native.lib_path())
    root = os.path.dirname(HERE)
    app = tmp_path / "app"
    shutil.copytree(os.path.join(root, "include"), app / "real_include")
    # compat/include -> <app>/include, compat/third_party -> <app>/third_party; the forwarders reach the binding through ../../ht_formats.hpp, i.e. <app>/ht_formats.hpp
    shutil.copytree(os.path.join(root, "include", "compat", "include"), app / "tree" / "include")
    shutil.copytree(os.path.join(root, "include", "compat", "third_party"), app / "tree" / "third_party")
    for f in os.listdir(app / "real_include"):
        if f.endswith((".hpp", ".h")):
            shutil.copy(app / "real_include" / f, app / f)
    os.makedirs(app / "tree" / "synthetic-hand-tracker")
    shutil.copy(os.path.join(HERE, "cxx_compat_shim.cpp"), app / "tree" / "synthetic-hand-tracker" / "synthetic-tracker-side.cpp")
    exe = str(tmp_path / "shim")
    subprocess.check_call(["g++", "-std=c++17", "-O1", str(app / "tree" / "synthetic-hand-tracker" / "synthetic-tracker-side.cpp"), "-o", exe, "-L" + lib, "-lht_mi355x", "-Wl,-rpath," + lib])
    assert "compiled and linked" in subprocess.check_output([exe]).decode()
    bank = "/root/reference/assets/animbank.pose"
    if os.path.exists(bank):
        assert "animbank rows=2336" in subprocess.check_output([exe, bank]).decode()


@pytest.mark.gpu
def test_overlapped_update_with_the_job_collected_first_equals_the_synchronous_update(tmp_path, golden, weights):
    """HandTracker::overlapped_update runs the CNN job on a second device context beside the caller's passes (the reference's std::async structure, handtrack.h:755-768).
    With overlapped_wait the job is waited for and collected before the passes: that IS the synchronous sequence (othermodel seeded from handmodel, the job, the accepted
    pose into handmodel, the passes), so every pose, heat-map and facade read must equal the synchronous tracker's bit for bit."""
    from hand_tracking_samples_amd import weights as W
    exe = _build(tmp_path)
    nf = 8
    depth = np.stack([golden["f%d/depth" % f] for f in range(nf)]); cams = np.stack([golden["f%d/cam" % f] for f in range(nf)])
    start = np.stack([golden["f%d/startpose" % f] for f in range(nf)]); gt = np.stack([golden["f%d/gtpose" % f] for f in range(nf)])
    _write_input(tmp_path / "in.bin", depth, cams, start, gt)
    cnnb = str(tmp_path / "w.cnnb")
    W.save_cnnb(cnnb, weights)
    outs = []
    for extra in ([], ["overlapped_wait"]):
        r = subprocess.run([exe, "track", ol.MODEL, cnnb, str(tmp_path / "in.bin"), str(tmp_path / ("out%d.bin" % len(outs)))] + extra, capture_output=True, text=True, timeout=300)
        print(r.stdout, r.stderr)
        assert r.returncode == 0 and "track: 8 frames, 17 bones" in r.stdout
        outs.append(np.fromfile(tmp_path / ("out%d.bin" % len(outs)), np.float32))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
def test_tracking_loop_through_the_cxx_surface_matches_the_reference(tmp_path, golden, weights):
    from hand_tracking_samples_amd import native, weights as W
    exe = _build(tmp_path)
    nf = 8
    depth = np.stack([golden["f%d/depth" % f] for f in range(nf)]); cams = np.stack([golden["f%d/cam" % f] for f in range(nf)])
    start = np.stack([golden["f%d/startpose" % f] for f in range(nf)]); gt = np.stack([golden["f%d/gtpose" % f] for f in range(nf)])
    _write_input(tmp_path / "in.bin", depth, cams, start, gt)
    cnnb = str(tmp_path / "w.cnnb")
    W.save_cnnb(cnnb, weights)
    r = subprocess.run([exe, "track", ol.MODEL, cnnb, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "track: 8 frames, 17 bones" in r.stdout
    rec = np.fromfile(tmp_path / "out.bin", np.float32).reshape(nf, 17 * 7 + 2304 + 2304 + 1 + 17 * 7)
    for f in range(nf):
        pose = rec[f, :119].reshape(17, 7); cnn = rec[f, 119:119 + 2304]; labels = rec[f, 119 + 2304:119 + 4608]; shown = rec[f, 119 + 4608]; facade = rec[f, 120 + 4608:].reshape(17, 7)
        ref = golden["f%d/uw_pose_user" % f]
        dp = np.abs(pose[:, :3] - ref[:, :3]).max(); dq = np.abs(pose[:, 3:] - ref[:, 3:]).max()
        print("C++ surface frame %d: |dpos| %.2e |dquat| %.2e vs the reference" % (f, dp, dq))
        assert dp <= POS_TOL and dq <= QUAT_TOL
        assert np.abs(cnn - golden["f%d/cnn_output" % f]).max() <= 2e-5
        assert np.array_equal(labels, native.expected_cnn(gt[f], cams[f]))      # GatherHandExpectedCNN through camsub(segment.cam, 4)
        assert shown == 1.0
        assert np.array_equal(facade, pose)                                      # handmodel.GetPoseUser() == what update() returned


def test_depth_mesh_and_heat_map_visualisation_match_the_reference(tmp_path):
    """host only: DepthMesh (misc_image.h:419-450) and VisualizeHMaps (handtrack.h:256-267) as synthetic-tracker.cpp:191,204-209 calls them, against what the
    reference produced for the same frame (tests/golden/viz1.npz from `ref_harness viz`): vertices, triangles and both label images byte for byte."""
    exe = _build(tmp_path)
    G = np.load(os.path.join(HERE, "golden", "viz1.npz"))
    depth = np.zeros((2, 240, 320), np.uint16); depth[0] = G["depth"]; depth[1].reshape(-1)[:4096] = G["tile"].reshape(-1)
    cams = np.stack([G["cam"], G["segcam"]]); pose = np.stack([G["pose"], G["pose"]])
    _write_input(tmp_path / "in.bin", depth, cams, pose, pose)
    subprocess.check_call([exe, "viz", "-", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = open(tmp_path / "out.bin", "rb").read()
    nv, nt, lw, lh, aw, ah = struct.unpack("<6i", raw[:24]); off = 24
    verts = np.frombuffer(raw, np.float32, nv * 3, off).reshape(nv, 3); off += nv * 12
    tris = np.frombuffer(raw, np.int32, nt * 3, off).reshape(nt, 3); off += nt * 12
    lab = np.frombuffer(raw, np.uint8, lw * lh * 3, off).reshape(lh, lw, 3); off += lw * lh * 3
    ang = np.frombuffer(raw, np.uint8, aw * ah * 3, off).reshape(ah, aw, 3)
    assert np.array_equal(verts, G["dm_verts"]) and np.array_equal(tris, G["dm_tris"])
    assert lab.shape == G["landmark_labels"].shape and np.array_equal(lab, G["landmark_labels"])
    assert np.array_equal(ang, G["angle_labels"])
    assert lab[..., 0].max() > 0 and (lab[..., 2] > 0).any()      # the heat-maps and the tile are both in the picture
