# Does an initialised RCCL process group (watchdog threads, extra streams) slow a step that does not use it?  One rank, no collective in the loop.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import bench
from hand_tracking_samples_amd import native, weights as W
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if mode != "none": dist.init_process_group("nccl", device_id=dev)
N = 1024
depth, cams, start = bench._load_frames(N)
c = native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), N)
c.load_weights(W.make_cnnb(W.DEFAULT_SEED, W.DEFAULT_FC2_GAIN)); c.set_params(microforce=3.0, mainthreadpasses=3)
b = (torch.from_numpy(depth.view(np.int16)).to(dev), torch.from_numpy(cams).to(dev), torch.from_numpy(start).to(dev), torch.empty((N, c.nb, 7), dtype=torch.float32, device=dev))
out = torch.empty((N, c.nb, 7), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream(dev)
def step():
    c.update_dev(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), N, b[3].data_ptr(), s.cuda_stream)
    if mode == "gather": dist.all_gather_into_tensor(out, b[3], async_op=True)
for _ in range(3): step()
torch.cuda.synchronize()
K = 10; t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("mode %s: %.3f ms/step, host issue %.3f ms/step" % (mode, (t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3), flush=True)
c.close()
if mode != "none": dist.destroy_process_group()
