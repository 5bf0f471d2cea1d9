import os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights
B = int(os.environ.get("FRAMES", "1024"))
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = np.arange(B) % 256
depth, cams, start = d["depth"][idx], d["cam"][idx], d["startpose"][idx]
ctx = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
ctx.load_weights(weights.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
ctx.tracker_reset(start)
ctx.stage_prepare(depth.reshape(B, 64, 64), cams)
e = ctx.stage_fit_error(0, B)
print("fit error mean", float(e.mean()))
