"""Bench input for BASELINE configs[4]: 64 frames of 128x128 depth of the 26-bone hand (every 36th animation-bank row), with the
next row as the carried pose.  Rendered by the reference's own model code in oracle/_ref/ref_harness:

    python tests/golden/make_model_hand26.py /tmp/model_hand26.json
    HT_REF_MODEL_JSON=/tmp/model_hand26.json oracle/_ref/ref_harness fullframes /root/reference/assets/animbank.pose 3 36 64 128,128,163 /tmp/frames5.htfx
    python tests/golden/make_frames5.py /tmp/frames5.htfx
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import htfx  # noqa: E402

G = htfx.load(sys.argv[1])
out = os.path.join(HERE, "frames5_64.npz")
np.savez_compressed(out, depth=G["depth"], cam=G["cam"], startpose=G["startpose"], rows=G["rows"])
print(out, os.path.getsize(out), G["depth"].shape, G["startpose"].shape)
