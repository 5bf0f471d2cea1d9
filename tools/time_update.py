"""Times ht_update_dev on the bench frames with some tracker parameters changed (upper bounds of what a phase costs a step):
    python tools/time_update.py [name=value ...]      e.g. full_reset_on_error=100 (no frame resets), physics_use_collision=0 (no contact kernel)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = int(os.environ.get("FRAMES", "1024"))
kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
d = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))      # the bench's 1024 distinct frames
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
dev = torch.device("cuda:0")
c = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0, mainthreadpasses=3, **kw)
if os.environ.get("BUILD"): c.debug_solver_build(int(os.environ["BUILD"]))      # pin k_solve's build: 1 small (8 frames per CU), 2 only (4 per CU), 3 mid
bufs = (torch.from_numpy(depth.view(np.int16)).to(dev), torch.from_numpy(cams).to(dev), torch.from_numpy(start).to(dev), torch.empty((B, 17, 7), dtype=torch.float32, device=dev))
s = torch.cuda.current_stream(dev)
def step(): c.update_dev(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), B, bufs[3].data_ptr(), s.cuda_stream)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%d frames, %s: %.3f ms/step  %.0f frames/s" % (B, kw or "defaults", dt * 1e3, B / dt), flush=True)
