"""Per-slab cycle account of k_fc and k_fc144 on a few waves (tuning build, HT_DEBUG_SKIP=0x800000): one CNN evaluation of 1024 frames.
k_fc144 at the time of writing: 240 k cycles for 64 slabs of 2304 matrix-pipe cycles; a wave with three loads stands ~1000 cycles per slab in the CU's address path."""
import os, sys
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights
B = 1024
ctx = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
ctx.load_weights(weights.make_cnnb())
x = np.random.default_rng(0).random((B, 4096), dtype=np.float32)
ctx.cnn_eval(x)
