"""How a rounding-level difference grows over the MultiStepSim steps of one sensitive case (golden frame 4 with othermodel started from frame 3's pose: tests/test_gpu_solver.py
test_update_cnn_model_and_kickstart): device against the restatement given the device's heat-maps, after 1 .. 5 steps.  A defect would show at the first step; an amplified
rounding difference starts at 1e-7 and grows from step to step.  With a -DHT_TUNING library (HT_LIB_PATH) HT_DEBUG_SKIP=65536 takes the single-body rows row by row instead of
four at a time: the other association order of the same sums."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import htfx, oracle_lib as ol
from hand_tracking_samples_amd import native, weights as W
g = htfx.load(os.path.join(ROOT, "tests", "golden", "golden8.htfx"))
NF = 8
depth = np.stack([g["f%d/depth" % f] for f in range(NF)]).reshape(NF, 64, 64); cams = np.stack([g["f%d/cam" % f] for f in range(NF)]); start = np.stack([g["f%d/startpose" % f] for f in range(NF)])
other = np.roll(start, 1, axis=0)
w = W.make_cnnb()
ctx = native.Context(ol.MODEL, NF); ctx.load_weights(w)
orc = ol.Oracle(w); orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
for steps in (1, 2, 3, 4, 5):
    ctx.set_params(microforce=3.0, mainthreadpasses=3, steps=steps)
    ctx.tracker_reset(start)
    st = ctx.get_state(1, NF); st[:, :, :7] = other; st[:, :, 7:] = 0.0; ctx.set_state(1, st)
    poses, acc = ctx.update_cnn_model_sync(depth, cams)
    cnn = ctx.cnn_results(NF)[1]
    orc.head.par.steps = steps
    out = []
    for f in range(NF):
        orc.reset(start[f]); so = orc.get_state(1); so[:, :7] = other[f]; orc.set_state(1, so)
        ref = np.zeros((17, 7), np.float32); cam = ol.camera(cams[f])
        y = np.ascontiguousarray(cnn[f]); orc.L.ho_set_cnn_override(orc.h, ol.fptr(y))
        orc.L.ho_update_cnn_model(orc.h, ol.u16ptr(np.ascontiguousarray(depth[f])), C.byref(cam), ol.fptr(ref))
        orc.L.ho_set_cnn_override(orc.h, None)
        ro = orc.get_state(1)
        out.append(max(np.abs(poses[f][:, :3] - ro[:, :3]).max(), np.abs(poses[f][:, 3:] - ro[:, 3:7]).max()))
    print("steps %d: max(|dpos|, |dquat|) per frame %s" % (steps, ["%.1e" % v for v in out]), flush=True)
