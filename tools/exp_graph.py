"""Experiment: one ht_update_dev step captured in a HIP graph (through torch.cuda.CUDAGraph) against the same step launched eagerly.
    python tools/exp_graph.py"""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = int(os.environ.get("FRAMES", "1024"))
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
dev = torch.device("cuda:0")
c = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0, mainthreadpasses=3)
bufs = (torch.from_numpy(depth.view(np.int16)).to(dev), torch.from_numpy(cams).to(dev), torch.from_numpy(start).to(dev), torch.empty((B, 17, 7), dtype=torch.float32, device=dev))
s = torch.cuda.Stream(dev)
def step(): c.update_dev(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), B, bufs[3].data_ptr(), s.cuda_stream)
def timed(fn, K=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
eager_out = bufs[3].clone()
print("eager: %.3f ms/step" % timed(step), flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph replay equals eager:", bool(torch.equal(bufs[3], eager_out)), flush=True)
    print("graph: %.3f ms/step" % timed(g.replay), flush=True)
    print("eager again: %.3f ms/step" % timed(step), flush=True)
except Exception as e:
    print("capture failed:", repr(e)[:500], flush=True)
