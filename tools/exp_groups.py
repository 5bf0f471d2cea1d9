# Experiment: the same 1024 frames served by G contexts of 1024/G frames each on their own streams (kernel boundaries then wait only for the
# slowest frame of a group).  usage: python tools/exp_groups.py [frames] [G ...]
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from hand_tracking_samples_amd import native, weights as W
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Gs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
depth, cams, start = bench._load_frames(N)
dev = torch.device("cuda", 0)
w = W.make_cnnb(W.DEFAULT_SEED, W.DEFAULT_FC2_GAIN)
for G in Gs:
    n = N // G
    ctxs, bufs, streams = [], [], []
    for g in range(G):
        c = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), n)
        c.load_weights(w); c.set_params(microforce=3.0, mainthreadpasses=3)
        sl = slice(g * n, (g + 1) * n)
        bufs.append((torch.from_numpy(depth[sl].view(np.int16)).to(dev), torch.from_numpy(cams[sl]).to(dev), torch.from_numpy(start[sl]).to(dev),
                     torch.empty((n, c.nb, 7), dtype=torch.float32, device=dev)))
        ctxs.append(c); streams.append(torch.cuda.Stream(dev))
    def step():
        for c, b, s in zip(ctxs, bufs, streams):
            c.update_dev(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), n, b[3].data_ptr(), s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("frames %d groups %d: %.3f ms/step (%.0f frames/s); host issue %.3f ms/step" % (N, G, (t2 - t0) / K * 1e3, N * K / (t2 - t0), (t1 - t0) / K * 1e3), flush=True)
    for c in ctxs: c.close()
