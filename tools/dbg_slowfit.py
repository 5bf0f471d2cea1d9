import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import htfx, oracle_lib as ol
from hand_tracking_samples_amd import native, weights as W
G = htfx.load(os.path.join(ROOT, "tests", "golden", "slowfit3.htfx"))
w = W.make_cnnb()
ctx = native.Context(ol.MODEL, 1); ctx.load_weights(w); ctx.set_params(microforce=3.0)
orc = ol.Oracle(w); orc.head.par.microforce = 3.0
f = 0
depth = G["f0/depth"].reshape(1, -1); cams = G["f0/cam"].reshape(1, 12); start = G["f0/startpose"][None]; crays = G["f0/crays"][None]
ctx.stage_prepare(depth, cams)
cam = ol.camera(G["f0/cam"]); buf = (ol.F3 * 4096)(); nfull = C.c_int()
n = orc.L.ho_pointcloud(ol.u16ptr(np.ascontiguousarray(depth[0])), C.byref(cam), 0.1, 0.7, 4, buf, 4096, C.byref(nfull))
for ncray in (1, 3, 4, 8):
  for steps in (1, 2, 6):
    ctx.tracker_reset(start); orc.reset(start[0])
    cr = np.zeros((1, 8, 4), np.float32); cr[0, :ncray] = crays[0, :ncray]
    ctx.L.ht_slowfit(ctx.h, 1, 0, None, steps, -1, None, None, native._f(np.ascontiguousarray(cr)), ncray)
    orc.L.ho_slowfit(orc.h, buf, n, 0, None, steps, -1, ol.F3(0, 0, 0), ol.F3(0, 0, 0), ol.fptr(np.ascontiguousarray(crays[0])), ncray)
    a = ctx.get_state(0, 1)[0]; b = orc.get_state(0)
    print("ncray", ncray, "steps", steps, "dpos %.2e dquat %.2e dmom %.2e" % (np.abs(a[:, :3] - b[:, :3]).max(), np.abs(a[:, 3:7] - b[:, 3:7]).max(), np.abs(a[:, 7:] - b[:, 7:]).max()))
