"""Diagnostic: config 5 end to end, device against the restatement per frame, with the decoded CNN analysis of both (which discrete decision flipped)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import htfx, oracle_lib as ol
from hand_tracking_samples_amd import native, weights as W
G = htfx.load(os.path.join(ROOT, "tests/golden/e2e128.htfx")); FR = np.load(os.path.join(ROOT, "tests/golden/frames5_64.npz"))
w = W.make_cnnb128(); n = len(FR["depth"])
ctx = native.Context(os.path.join(ROOT, "tests/golden/model_hand26.htfx"), n)
ctx.load_weights128(w); ctx.set_params(microforce=3.0, mainthreadpasses=3); ctx.tracker_reset(FR["startpose"])
poses, cnn = ctx.update_direct_sync(FR["depth"], FR["cam"], 128, want_cnn=True)
_, cnn_dev, an_dev = ctx.cnn_results(n)
other = ctx.get_state(1, n)[:, :, :7]
do = np.abs(other - G["all/other_pose"]).max(axis=(1, 2))
dp = np.abs(poses - G["all/uw_pose_user"]).max(axis=(1, 2))
L = ol.lib()
for i in range(n):
    x = np.zeros(128 * 128, np.float32)
    L.ho_cnn_input(ol.u16ptr(np.ascontiguousarray(FR["depth"][i]).reshape(-1)), 128 * 128, float(FR["cam"][i][4]), 0.1, 0.7, ol.fptr(x))
    y = ol.cnn128_eval(w, x[None])[0]
    c = FR["cam"][i].copy(); c[:4] /= 8.0
    hcam = ol.camera(c, 16, 16); an = ol.Analysis()
    L.ho_decode(ol.fptr(y), C.byref(hcam), C.byref(an))
    ip = np.array([[p.x, p.y] for p in an.image_points], np.float32); vals = np.array(list(an.vals), np.float32)
    dip = np.abs(an_dev[i, 32:48].reshape(8, 2) - ip).max(); dv = np.abs(an_dev[i, 56:72] - vals).max()
    print("frame %2d: other %.2e user %.2e | dcnn %.2e  d(image_points) %.2e  d(vals) %.2e" % (i, do[i], dp[i], np.abs(cnn_dev[i] - y).max(), dip, dv))
ctx.close()
# the restatement's tracker fed with the DEVICE's CNN output: what is left is the solver's own difference
orc = ol.Oracle(None, model=os.path.join(ROOT, "tests/golden/model_hand26.htfx"))
assert L.ho_set_direct(orc.h, 128, ol.fptr(w), w.size) == 0
orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
for i in range(n):
    y = np.ascontiguousarray(cnn_dev[i]); L.ho_set_cnn_override(orc.h, ol.fptr(y))
    orc.reset(FR["startpose"][i]); user = np.zeros((26, 7), np.float32); cam = ol.camera(FR["cam"][i], 128, 128)
    L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(FR["depth"][i])), C.byref(cam), ol.fptr(user))
    o = orc.get_state(1)[:, :7]
    print("frame %2d with the device's CNN output: other %.2e (pos %.2e) user pos %.2e quat %.2e" % (i, np.abs(o - other[i]).max(), np.abs(o[:, :3] - other[i][:, :3]).max(), np.abs(user[:, :3] - poses[i][:, :3]).max(), np.abs(user[:, 3:] - poses[i][:, 3:]).max()))
L.ho_set_cnn_override(orc.h, None)
