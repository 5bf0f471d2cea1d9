"""Times ht_update_dev on the bench frames for a given full_reset_on_error (0 = every frame takes the reset path)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = 1024
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
d = np.load(os.path.join(ROOT, "tests", "golden", "frames256.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
dev = torch.device("cuda:0")
c = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0, mainthreadpasses=3, full_reset_on_error=thr)
bufs = (torch.from_numpy(depth.view(np.int16)).to(dev), torch.from_numpy(cams).to(dev), torch.from_numpy(start).to(dev), torch.empty((B, 17, 7), dtype=torch.float32, device=dev))
s = torch.cuda.current_stream(dev)
def step(): c.update_dev(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), B, bufs[3].data_ptr(), s.cuda_stream)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("full_reset_on_error %.2f: %.3f ms/step  %.0f frames/s" % (thr, dt * 1e3, B / dt), flush=True)
