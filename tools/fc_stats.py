"""Per-slab cycle account of k_fc and k_fc144_pk on a few waves (tuning build, HT_DEBUG_SKIP=8388608 = 0x800000, decimal: the switch is read with atoi): one CNN evaluation of 1024 frames.
The last layer over the rounds: 240 k cycles for 64 slabs of 2304 matrix-pipe cycles with register staging (a wave with three loads stood ~1000 cycles per slab in the CU's address
path), 218 k with the tiles by LDS-DMA (3410 per slab: 140 own pieces, 150 barrier, 415 DMA issue, 2710 reads + matrix instructions on the wave that ends the slab)."""
import os, sys
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights
B = 1024
ctx = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
ctx.load_weights(weights.make_cnnb())
x = np.random.default_rng(0).random((B, 4096), dtype=np.float32)
ctx.cnn_eval(x)
