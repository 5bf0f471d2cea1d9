import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import htfx, oracle_lib as ol
import test_gpu_solver as T
from hand_tracking_samples_amd import native, weights as W
g = htfx.load(os.path.join(ROOT, "tests/golden/golden8.htfx"))
NF = 8
ctx = native.Context(ol.MODEL, NF); ctx.load_weights(W.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3); ctx.debug_solver_build(5)
depth, cams, start = T._inputs(g)
for steps in (1, 2, 3, 5):
    ctx.stage_prepare(depth, cams); ctx.tracker_reset(start); ctx.set_params(steps=steps)
    ctx.stage_multistep(T._analysis(g), NF)
    got = ctx.get_state(1, NF)
    print("steps %d:" % steps, " ".join("%.1e" % np.abs(got[f] - g["f%d/multistep%d" % (f, steps)]).max() for f in range(NF)))
ctx.set_params(steps=5)
for k in range(4):
    ctx.stage_prepare(depth, cams); ctx.tracker_reset(start)
    ctx.stage_scratch_unibody(T._analysis(g), NF, k)
    got = ctx.get_state(1, NF)
    print("reset k=%d:" % k, " ".join("%.1e" % np.abs(got[f][:, :7] - g["f%d/%s" % (f, "scratch" if k == 0 else "unibody%d" % (k - 1))][:, :7]).max() for f in range(3)))
