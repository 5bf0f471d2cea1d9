"""Diagnostic: the exact-order solver build against the restatement fed with the device's CNN output; where the first difference enters."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from hand_tracking_samples_amd import native, weights as W
FR = np.load(os.path.join(ROOT, "tests/golden/frames256.npz")); N = 64
w = W.make_cnnb()
ctx = native.Context(ol.MODEL, N); ctx.load_weights(w); ctx.set_params(microforce=3.0, mainthreadpasses=3); ctx.debug_solver_build(5)
ctx.tracker_reset(FR["startpose"][:N])
poses, cnn = ctx.update_sync(FR["depth"][:N].reshape(N, -1), FR["cam"][:N], want_cnn=True)
other = ctx.get_state(1, N); _, _, an_dev = ctx.cnn_results(N)
L = ol.lib()
orc = ol.Oracle(w); orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
for i in range(N):
    y = np.ascontiguousarray(cnn[i]); L.ho_set_cnn_override(orc.h, ol.fptr(y))
    orc.reset(FR["startpose"][i]); user = np.zeros((17, 7), np.float32); cam = ol.camera(FR["cam"][i])
    L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(FR["depth"][i]).reshape(-1)), C.byref(cam), ol.fptr(user))
    o = orc.get_state(1)
    c = FR["cam"][i].copy(); c[:4] /= 4.0
    hcam = ol.camera(c, 16, 16); an = ol.Analysis(); L.ho_decode(ol.fptr(y), C.byref(hcam), C.byref(an))
    flat = np.frombuffer(bytes(an), np.float32)
    da = np.abs(an_dev[i] - flat[:84]); k = int(np.argmax(da))
    print("frame %2d: other %.2e user %.2e | analysis max diff %.2e at index %d" % (i, np.abs(o - other[i]).max(), np.abs(user - poses[i]).max(), da.max(), k))
