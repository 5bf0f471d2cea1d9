import os, sys
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights
B = 256
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
depth, cams, start = d["depth"][:B], d["cam"][:B], d["startpose"][:B]
ctx = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
ctx.load_weights(weights.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
ctx.tracker_reset(start)
ctx.stage_prepare(depth.reshape(B, 64, 64), cams)
rows, n = ctx.stage_cloud_rows(0, 1, 0, B)
n = np.asarray(n)
want = n // 1000; pts = n % 1000
print("points/frame mean %.0f; facing away mean %.1f (%.1f %%), p90 %.0f, max %d" % (pts.mean(), want.mean(), 100.0 * want.sum() / pts.sum(), np.percentile(want, 90), want.max()))
