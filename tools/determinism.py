import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from hand_tracking_samples_amd import native, weights as W
ROOT = "/root/repo"
for B in (1024, 1536, 2048, 37):
    depth, cams, start = bench._load_frames(B)
    c = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
    c.load_weights(W.make_cnnb(W.DEFAULT_SEED, W.DEFAULT_FC2_GAIN)); c.set_params(microforce=3.0, mainthreadpasses=3)
    ref = None; bad = 0
    for it in range(6):
        c.tracker_reset(start)
        p = c.update_sync(depth.reshape(B, 64, 64), cams)
        p2 = c.update_sync(depth.reshape(B, 64, 64), cams)      # streaming second frame
        key = np.concatenate([p.ravel(), p2.ravel()])
        if ref is None: ref = key
        elif not np.array_equal(ref, key): bad += 1
    print("B", B, "repeats differing from the first:", bad, "capacity", c.capacity_events(), flush=True)
    c.close()
