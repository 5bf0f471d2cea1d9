"""Experiment: the 1024-frame step as G independent sub-batches (one context and stream each) running side by side on the GPU.
    python tools/exp_split_batch.py [G ...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from hand_tracking_samples_amd import native, weights as W
B = int(os.environ.get("FRAMES", "1024"))
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
dev = torch.device("cuda:0")
w = W.make_cnnb()
for G in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    n = B // G
    ctxs, bufs, streams = [], [], []
    for g in range(G):
        c = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), n)
        c.load_weights(w); c.set_params(microforce=3.0, mainthreadpasses=3)
        sl = slice(g * n, (g + 1) * n)
        bufs.append((torch.from_numpy(depth[sl].view(np.int16)).to(dev), torch.from_numpy(cams[sl]).to(dev), torch.from_numpy(start[sl]).to(dev), torch.empty((n, 17, 7), dtype=torch.float32, device=dev)))
        ctxs.append(c); streams.append(torch.cuda.Stream(dev))
    def step():
        for c, b, s in zip(ctxs, bufs, streams):
            c.update_dev(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), n, b[3].data_ptr(), s.cuda_stream if os.environ.get("TORCH_STREAMS") else 0)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 10
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%d frames as %d x %d: %.3f ms/step  %.0f frames/s" % (B, G, n, dt * 1e3, B / dt), flush=True)
    del ctxs, bufs
