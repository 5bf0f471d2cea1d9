"""Host-resident input without a copy engine in the step's way: the depth frames stay in PINNED host memory and k_prepare reads them over the host link itself (a pinned
allocation has one address for host and device); the cameras (48 KB, read by many kernels) are copied; the poses are written to pinned host memory by the kernel that makes them.
Against bench.py's host_io leg (uploads on a copy stream beside the step) and the resident loop, same pacing.   FRAMES=1024 python tools/exp_hostio_zero_copy.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = int(os.environ.get("FRAMES", "1024")); N = int(os.environ.get("STEPS", "30"))
dev = torch.device("cuda", 0)
z = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
idx = np.arange(B) % 1024
depth, cams, start = z["depth"].reshape(-1, 4096)[idx].astype(np.uint16), z["cam"][idx].astype(np.float32), z["startpose"][idx].astype(np.float32)
ctx = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
ctx.load_weights(W.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
d_start = torch.from_numpy(start).to(dev)
h_depth = [torch.from_numpy(depth.view(np.int16)).pin_memory() for _ in range(2)]; h_cams = torch.from_numpy(cams).pin_memory()
h_poses = [torch.empty((B, 17, 7), dtype=torch.float32).pin_memory() for _ in range(2)]
din = [h_depth[0].to(dev) for _ in range(2)]; cin = [h_cams.to(dev) for _ in range(2)]
dout = [torch.empty((B, 17, 7), dtype=torch.float32, device=dev) for _ in range(2)]
stream = torch.cuda.current_stream(dev); copy = torch.cuda.Stream(device=dev)
e_in = [torch.cuda.Event() for _ in range(2)]; e_out = [torch.cuda.Event() for _ in range(2)]; e_used = [torch.cuda.Event() for _ in range(2)]


def run(n, mode):
    for b in range(2):
        e_used[b].record(stream)
    def upload(k):
        with torch.cuda.stream(copy):
            e_used[k % 2].synchronize()
            if mode == "copy":
                din[k % 2].copy_(h_depth[k % 2], non_blocking=True)
            cin[k % 2].copy_(h_cams, non_blocking=True)
            e_in[k % 2].record(copy)
    if mode != "resident":
        upload(0)
    for k in range(n):
        if mode != "resident" and k + 1 < n:
            upload(k + 1)
        elif mode == "resident":
            e_used[(k + 1) % 2].synchronize()
        if mode != "resident":
            stream.wait_event(e_in[k % 2])
        src = h_depth[k % 2].data_ptr() if mode in ("zero", "zero_in") else din[k % 2].data_ptr()
        dst = h_poses[k % 2].data_ptr() if mode == "zero" else dout[k % 2].data_ptr()
        ctx.update_dev(src, cin[k % 2].data_ptr(), d_start.data_ptr(), B, dst, stream.cuda_stream)
        e_out[k % 2].record(stream); e_used[k % 2].record(stream)
        if mode in ("copy", "zero_in"):
            with torch.cuda.stream(copy):
                copy.wait_event(e_out[k % 2])
                h_poses[k % 2].copy_(dout[k % 2], non_blocking=True)
    torch.cuda.synchronize()


ref = None
for mode in ("resident", "copy", "zero_in", "zero", "resident", "copy", "zero_in", "zero"):
    run(3, mode)
    t0 = time.perf_counter(); run(N, mode); dt = (time.perf_counter() - t0) / N * 1e3
    p = h_poses[(N - 1) % 2].clone() if mode != "resident" else dout[(N - 1) % 2].cpu()
    if ref is None: ref = p
    print("%-9s %.3f ms per step, %.0f frames/s, poses equal the resident run: %s" % (mode, dt, B / dt * 1e3, bool(torch.equal(p, ref))), flush=True)
ctx.close()
