import os, time, sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
a = torch.randn(1024, 17, 7, device=dev); out = torch.empty_like(a)
for asyn in (False, True):
    for _ in range(3): w = dist.all_gather_into_tensor(out, a, async_op=asyn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): w = dist.all_gather_into_tensor(out, a, async_op=asyn)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("async", asyn, "host per call %.3f ms, total per call %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
# with a busy GPU: a long kernel first
x = torch.randn(8192, 8192, device=dev)
for asyn in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        y = x @ x
        w = dist.all_gather_into_tensor(out, a, async_op=asyn)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("busy async", asyn, "host %.3f ms total %.3f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): y = x @ x
torch.cuda.synchronize(); print("matmul alone %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
dist.destroy_process_group()
