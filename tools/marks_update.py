"""Event marks of one update on the tuning build (HT_MARKS=1): where the concurrent streams really are, in ms since the fork."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W
B = int(os.environ.get("FRAMES", "1024"))
d = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx].reshape(B, -1), d["cam"][idx], d["startpose"][idx]
dev = torch.device("cuda:0")
c = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
c.load_weights(W.make_cnnb()); c.set_params(microforce=3.0, mainthreadpasses=3)
bufs = (torch.from_numpy(depth.view(np.int16)).to(dev), torch.from_numpy(cams).to(dev), torch.from_numpy(start).to(dev), torch.empty((B, 17, 7), dtype=torch.float32, device=dev))
s = torch.cuda.current_stream(dev)
for i in range(int(os.environ.get("N", "4"))):
    sys.stderr.write("update %d\n" % i)
    c.update_dev(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), B, bufs[3].data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
