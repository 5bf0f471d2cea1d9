"""How far the product solver is from the restatement after 1 .. 5 free-running MultiStepSim steps, on the bench frames whose CNN-driven pose leaves the tight band
(profiles/r05_gpu_tests.log: othermodel frames 6, 130, 217, 939; frame 212 with always_take_cnn), the restatement given the device's own heat-maps.  Beside it the same
frames' SINGLE steps from the restatement's state (what tests/test_gpu_teacher_forced.py asserts on all 1024 frames).  A defect shows at the first step at full size; a
rounding difference next to a discrete decision starts at 1e-7 and is multiplied from step to step.

    python tools/diag_growth.py [frame ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol, parity_rule as pr
from hand_tracking_samples_amd import native, weights as W
F = [int(a) for a in sys.argv[1:]] or [6, 130, 217, 939, 212]
d = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
depth, cams, start = d["depth"][F].reshape(len(F), -1), d["cam"][F], d["startpose"][F]
w = W.make_cnnb()
n = len(F)
ctx = native.Context(ol.MODEL, n); ctx.load_weights(w)
free = []
for steps in range(1, 6):
    ctx.set_params(microforce=3.0, mainthreadpasses=3, steps=steps)
    ctx.tracker_reset(start)
    ctx.update_cnn_model_sync(depth.reshape(n, 64, 64), cams)
    free.append(ctx.get_state(1, n))
cnn = ctx.cnn_results(n)[1]
orc = ol.Oracle(w); orc.head.par.microforce = 3.0; orc.head.par.mainthreadpasses = 3
trace = np.zeros((n, 10, 17, 13), np.float32); an = np.zeros((n, 84), np.float32); user = np.zeros((17, 7), np.float32)
for i in range(n):
    orc.reset(start[i]); orc.L.ho_set_trace(orc.h, ol.fptr(trace[i]))
    y = np.ascontiguousarray(cnn[i]); orc.L.ho_set_cnn_override(orc.h, ol.fptr(y))
    cam = ol.camera(cams[i])
    orc.L.ho_update(orc.h, ol.u16ptr(np.ascontiguousarray(depth[i])), C.byref(cam), ol.fptr(user))
    orc.L.ho_get_analysis(orc.h, ol.fptr(an[i]))
orc.L.ho_set_cnn_override(orc.h, None); orc.L.ho_set_trace(orc.h, None)
ctx.set_params(microforce=3.0, mainthreadpasses=3, steps=5)
ctx.stage_prepare(depth, cams)
single = []
for s in range(5):
    ctx.set_state(1, trace[:, s]); ctx.stage_multistep_range(an, n, s, s + 1); single.append(ctx.get_state(1, n))
print("| frame | " + " | ".join("after %d steps, free-running" % k for k in range(1, 6)) + " | " + " | ".join("step %d alone" % s for s in range(5)) + " |")
print("|---|" + "---|" * 10)
for i, f in enumerate(F):
    row = []
    for k in range(5):
        dp, dq = pr.pose_diff(free[k][i:i + 1, :, :7], trace[i:i + 1, k + 1, :, :7]); row.append("%.1e m / %.1e" % (dp[0], dq[0]))
    for s in range(5):
        dp, dq = pr.pose_diff(single[s][i:i + 1, :, :7], trace[i:i + 1, s + 1, :, :7]); row.append("%.1e / %.1e" % (dp[0], dq[0]))
    print("| %d | %s |" % (f, " | ".join(row)))
