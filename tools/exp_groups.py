"""Experiment: the 1024-frame step split into G independent sub-batches, each on its own context and stream."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = 1024
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
dev = torch.device("cuda:0")
w = W.make_cnnb()
for G in (1, 2, 4, 8):
    n = B // G
    ctxs, bufs, streams = [], [], []
    for g in range(G):
        c = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), n)
        c.load_weights(w); c.set_params(microforce=3.0, mainthreadpasses=3)
        sl = slice(g * n, (g + 1) * n)
        bufs.append((torch.from_numpy(depth[sl].view(np.int16)).to(dev), torch.from_numpy(cams[sl]).to(dev), torch.from_numpy(start[sl]).to(dev), torch.empty((n, 17, 7), dtype=torch.float32, device=dev)))
        ctxs.append(c); streams.append(torch.cuda.Stream(dev))
    def step():
        for c, b, s in zip(ctxs, bufs, streams):
            c.update_dev(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), n, b[3].data_ptr(), s.cuda_stream)
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 6
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("groups %d: %.3f ms/step  %.0f frames/s" % (G, dt * 1e3, B / dt), flush=True)
    for c in ctxs: c.close()
