"""Worst case of the full-reset branch: EVERY frame of a 1024-frame batch takes PoseFromScratch + 3 x UnibodyFit (k_reset; the first update of a sequence does)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = 1024
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams = d["depth"][idx].reshape(B, -1), d["cam"][idx]
c = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0)
cnn_in, pts, n = c.stage_prepare(depth, cams)
an = c.stage_decode(c.cnn_eval(cnn_in), cams)
for _ in range(3): c.stage_scratch_unibody(an, B, 3)
t0 = time.perf_counter()
for _ in range(10): c.stage_scratch_unibody(an, B, 3)
print("all %d frames through the reset branch: %.3f ms" % (B, (time.perf_counter() - t0) / 10 * 1e3))
