"""bench.py's host_io leg with event timing: how long the upload takes while a step runs beside it, and how long the step takes with an upload beside it"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hand_tracking_samples_amd import native, weights as W
B = int(os.environ.get("FRAMES", "1024"))
dev = torch.device("cuda", 0)
z = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
idx = np.arange(B) % 1024
depth, cams, start = z["depth"].reshape(-1, 4096)[idx].astype(np.uint16), z["cam"][idx].astype(np.float32), z["startpose"][idx].astype(np.float32)
ctx = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
ctx.load_weights(W.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
d_start = torch.from_numpy(start).to(dev)
h_depth = torch.from_numpy(depth.view(np.int16)).pin_memory(); h_cams = torch.from_numpy(cams).pin_memory()
din = [h_depth.to(dev) for _ in range(2)]; cin = [h_cams.to(dev) for _ in range(2)]
dout = [torch.empty((B, 17, 7), dtype=torch.float32, device=dev) for _ in range(2)]
stream = torch.cuda.current_stream(dev); copy = torch.cuda.Stream(device=dev)
def run(n, h2d):
    c0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; c1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    s0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; s1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for k in range(n):
        if k > 0: s1[k - 1].synchronize() if k > 1 else None
        if h2d:
            with torch.cuda.stream(copy):
                c0[k].record(copy); din[(k + 1) % 2].copy_(h_depth, non_blocking=True); cin[(k + 1) % 2].copy_(h_cams, non_blocking=True); c1[k].record(copy)
        s0[k].record(stream)
        ctx.update_dev(din[k % 2].data_ptr(), cin[k % 2].data_ptr(), d_start.data_ptr(), B, dout[k % 2].data_ptr(), stream.cuda_stream)
        s1[k].record(stream)
    torch.cuda.synchronize()
    steps = [s0[k].elapsed_time(s1[k]) for k in range(3, n)]
    copies = [c0[k].elapsed_time(c1[k]) for k in range(3, n)] if h2d else [0.0]
    return np.median(steps), np.median(copies)
for h2d in (False, True, False, True):
    st, cp = run(16, h2d)
    print("upload beside the step: %-5s  step %.3f ms, upload %.3f ms" % (h2d, st, cp))
ctx.close()
