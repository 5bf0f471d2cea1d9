"""A TRAINED net (the reference's handposedd.cnnb is not shipped; section 8f next-2: "regenerate a trained .cnnb ... exercise crisp-peak decode").

tools/train_synthetic.py trains the 64x64 net with the repo's own training step (ht_cnn_train = CNN::Train, cnn.h:558-580) as train-hand-pose-cnn does
(train-cnn.cpp:156-162: batch-1 SGD, lr 0.001, labels = GatherHandExpectedCNN of the frame's ground-truth pose, handtrack.h:160-173) from seeded Xavier weights with FC2
gain 1 -- no artificial gain -- on 960 of the bench's 1024 software-rendered tiles; every 16th tile is held out.  ~290 k steps, ~17 s on the GPU; seed and script are
committed, the weights are not.  With it the CNN half of the path matters without a switch: the heat-maps have real peaks where the hand's landmarks are, the decode
(CNNOutputAnalysis, handtrack.h:182-242) works on them, and the tracker's accept rule (handtrack.h:720-722) takes the CNN-driven pose where it explains the frame better."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
EPOCHS = 300


@pytest.fixture(scope="module")
def trained():
    import train_synthetic as ts
    r = ts.train(EPOCHS)
    c = r["curve"]
    print("trained: %d steps in %.0f s; train mse %.3e -> %.3e, held-out %.3e -> %.3e" % (c[-1]["steps"], r["seconds"], c[0]["train_mse"], c[-1]["train_mse"], r["held_out_mse_before"], c[-1]["held_out_mse"]))
    return r


def test_the_loss_falls_tenfold_and_the_held_out_heat_maps_have_peaks_where_the_labels_do(trained):
    c = trained["curve"]
    assert c[-1]["train_mse"] * 10 <= c[0]["train_mse"]
    assert c[-1]["held_out_mse"] * 10 <= trained["held_out_mse_before"]
    y, t = trained["test_outputs"], trained["labels"][trained["test_frames"]]
    hm, lb = y[:, :2048].reshape(-1, 8, 16, 16), t[:, :2048].reshape(-1, 8, 16, 16)
    peak = hm.reshape(-1, 8, 256).max(axis=2)
    am, al = hm.reshape(-1, 8, 256).argmax(axis=2), lb.reshape(-1, 8, 256).argmax(axis=2)
    dist = np.hypot(am % 16 - al % 16, am // 16 - al // 16)
    print("held-out landmark heat-maps: mean peak %.3f (uniform 0.0039), peak within 1 cell of the label's on %.0f %% of the maps, mean distance %.2f cells" % (peak.mean(), 100 * (dist <= 1.5).mean(), dist.mean()))
    assert peak.mean() >= 0.1 and (dist <= 1.5).mean() >= 0.8


def test_net_and_decode_with_trained_weights_against_the_restatement(trained):
    """the device's forward pass and decode on the 64 held-out tiles with the trained weights (FC2 gain 1) against the restatement's: heat-maps to the CNN tolerance of
    tests/test_gpu_cnn.py, the decode of the device's own heat-maps as tests/test_gpu_cnn.py holds it on the golden frames"""
    from hand_tracking_samples_amd import native
    w = trained["weights"]; fr = trained["test_frames"]; x = trained["inputs"][fr]
    d = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))
    ctx = native.Context(ol.MODEL, len(fr))
    try:
        ctx.load_weights(w)
        y = ctx.cnn_eval(x)
        an = ctx.stage_decode(y, d["cam"][fr])
    finally:
        ctx.close()
    L = ol.lib()
    yr = np.zeros_like(y)
    for i in range(len(fr)):
        L.ho_cnn_eval(ol.fptr(w), ol.fptr(np.ascontiguousarray(x[i])), ol.fptr(yr[i]), None)
    print("trained net, 64 held-out tiles: |device - restatement| max %.2e (largest output %.3f)" % (np.abs(y - yr).max(), y.max()))
    assert np.abs(y - yr).max() <= 2e-5
    ar = np.zeros((len(fr), 84), np.float32)
    for i in range(len(fr)):
        hc = ol.camera(d["cam"][fr[i]], 16, 16)      # camsub(cam, 4), misc_image.h:60
        hc.focal.x /= 4.0; hc.focal.y /= 4.0; hc.principal.x /= 4.0; hc.principal.y /= 4.0
        a = ol.Analysis()
        L.ho_decode(ol.fptr(np.ascontiguousarray(y[i])), C.byref(hc), C.byref(a))
        ar[i] = np.frombuffer(bytes(a), np.float32)
    # every stage of the decode but sinf / cosf is the same IEEE arithmetic (tests/test_gpu_cnn.py: test_decode_matches_golden): rays, image points, confidences, the sixteen
    # angle bins and the clench angles bit for bit; the three palm angles and palmq through glibc's float sine to 2e-7
    assert np.array_equal(an[:, 0:72], ar[:, 0:72]) and np.array_equal(an[:, 79:84], ar[:, 79:84])
    assert np.allclose(an[:, 72:79], ar[:, 72:79], rtol=0, atol=2e-7)


def test_unit_of_work_with_the_trained_net_takes_the_cnn_pose_where_the_reference_does(trained):
    """the whole unit of work on the held-out tiles with the trained net, device against the restatement given the device's heat-maps: the accept decision (handtrack.h:720-722)
    frame by frame, the tracker flags, the poses inside the bands of tests/parity_rule.py"""
    from hand_tracking_samples_amd import native
    import parity_rule as pr
    w = trained["weights"]; fr = trained["test_frames"]
    d = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))
    depth, cams, start = d["depth"][fr].reshape(len(fr), -1), d["cam"][fr], d["startpose"][fr]
    n = len(fr)
    ctx = native.Context(ol.MODEL, n)
    try:
        ctx.load_weights(w)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(start)
        _, acc = ctx.update_cnn_model_sync(depth.reshape(n, 64, 64), cams)      # the CNN job alone: which frames would take the CNN-driven pose
        ctx.tracker_reset(start)
        poses, cnn = ctx.update_sync(depth, cams, want_cnn=True)
        pfe, ini = ctx.tracker_flags(n)
        other = ctx.get_state(1, n)[:, :, :7]
    finally:
        ctx.close()
    orc = ol.Oracle(w)
    orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
    user = np.zeros((n, 17, 7), np.float32); oth = np.zeros((n, 17, 7), np.float32); acc_r = np.zeros(n, np.int32); ini_r = np.zeros(n, np.int32)
    try:
        for i in range(n):
            cam = ol.camera(cams[i])
            y = np.ascontiguousarray(cnn[i]); orc.L.ho_set_cnn_override(orc.h, ol.fptr(y))
            orc.reset(start[i])
            acc_r[i] = orc.L.ho_update_cnn_model(orc.h, ol.u16ptr(np.ascontiguousarray(depth[i])), C.byref(cam), ol.fptr(user[i]))      # 0 or nb
            orc.reset(start[i])
            orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[i])), C.byref(cam), ol.fptr(user[i]))
            oth[i] = orc.get_state(1)[:, :7]; ini_r[i] = orc.flags()[1]
        orc.L.ho_set_cnn_override(orc.h, None)
    finally:
        orc.close()
    dp, dq = pr.pose_diff(poses, user); do = np.maximum(*pr.pose_diff(other, oth))
    print("trained net, unit of work on 64 held-out tiles: CNN pose accepted on %d (restatement: %d); user poses |dpos| max %.2e |dquat| max %.2e, %d inside 2e-5 m / 2e-4; othermodel median %.1e max %.1e"
          % (int(acc.sum()), int((acc_r > 0).sum()), dp.max(), dq.max(), int(((dp <= pr.TIGHT[0]) & (dq <= pr.TIGHT[1])).sum()), np.median(do), do.max()))
    assert np.array_equal(acc.astype(bool), acc_r > 0)
    assert np.array_equal(ini, ini_r)
    assert acc.sum() * 2 > n      # the accept branch fires on a majority of these frames
    assert np.median(do) <= pr.LOOSE[0] and (dp <= pr.CAP[0]).all() and (dq <= pr.CAP[1]).all() and ((dp <= pr.LOOSE[0]) & (dq <= pr.LOOSE[1])).sum() >= n - 4
