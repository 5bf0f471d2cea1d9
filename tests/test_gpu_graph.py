"""An update has to survive a HIP graph capture (every stream it forks comes back to the caller's, no fork out of a forked stream -- which the runtime of this
image does not survive): ht_update_dev captured once through torch.cuda.CUDAGraph and replayed gives the poses of the eager call bit for bit.  Run in a child
process: a capture that goes wrong dies inside the runtime, not with a Python exception."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from hand_tracking_samples_amd import native, weights as W
B = 512
d = np.load(os.path.join(%(root)r, "tests", "golden", "frames256.npz"))
idx = np.arange(B) %% len(d["depth"])
dev = torch.device("cuda:0")
c = native.Context(os.path.join(%(root)r, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0, mainthreadpasses=3)
bufs = (torch.from_numpy(d["depth"][idx].reshape(B, -1).view(np.int16)).to(dev), torch.from_numpy(d["cam"][idx]).to(dev), torch.from_numpy(d["startpose"][idx]).to(dev),
        torch.empty((B, 17, 7), dtype=torch.float32, device=dev))
s = torch.cuda.Stream(dev)
def step(): c.update_dev(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), B, bufs[3].data_ptr(), s.cuda_stream)
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.synchronize()
eager = bufs[3].clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
torch.cuda.synchronize()
bufs[3].zero_()
g.replay(); torch.cuda.synchronize()
assert bool(torch.isfinite(eager).all())
assert torch.equal(bufs[3], eager), "replay differs from the eager update"
g.replay(); torch.cuda.synchronize()
assert torch.equal(bufs[3], eager)
print("GRAPH-OK")
'''


def test_update_survives_graph_capture():
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH-OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
