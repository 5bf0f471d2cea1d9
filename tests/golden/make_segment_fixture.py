"""tests/golden/segment6.npz: six 320x240 depth frames rendered from the reference's animation bank and what the reference's
HandSegmentVR (handtrack.h:280-344) returns for them.  Regenerate in the build container with

    oracle/_ref/ref_harness segment /root/reference/assets/animbank.pose 0,144,912,1504,2048,2224 /tmp/seg.htfx
    python tests/golden/make_segment_fixture.py /tmp/seg.htfx
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htfx  # noqa: E402

d = htfx.load(sys.argv[1])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "segment6.npz")
np.savez_compressed(out, **{k.replace("/", "__"): v for k, v in d.items()})
print(out, os.path.getsize(out), "bytes,", len(d), "arrays")
