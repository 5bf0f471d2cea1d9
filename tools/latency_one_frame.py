"""Single-frame latency of every kernel of the update: run under `rocprofv3 --kernel-trace --stats` with B frames (default 1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = (first + np.arange(B)) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
c = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0, mainthreadpasses=3)
for _ in range(5):
    c.tracker_reset(start)
    c.update_sync(depth, cams)
c.close()
