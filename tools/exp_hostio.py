"""What the host transfers of a step cost on their own (pinned host memory, one copy stream): tools aid behind bench.py's host_io leg."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda", 0)
for B in (1024, 8192):
    h = torch.empty((B, 4096), dtype=torch.int16).pin_memory(); d = torch.empty((B, 4096), dtype=torch.int16, device=dev)
    hp = torch.empty((B, 17, 7), dtype=torch.float32).pin_memory(); dp = torch.empty((B, 17, 7), dtype=torch.float32, device=dev)
    s = torch.cuda.Stream(device=dev)
    for name, fn in (("H2D depth", lambda: d.copy_(h, non_blocking=True)), ("D2H poses", lambda: hp.copy_(dp, non_blocking=True))):
        with torch.cuda.stream(s):
            for _ in range(3): fn()
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            s.synchronize(); dt = (time.perf_counter() - t0) / 20
        nbytes = (h.numel() * 2) if name[0] == "H" else hp.numel() * 4
        print("%d frames: %s %.3f ms (%.1f GB/s)" % (B, name, dt * 1e3, nbytes / dt / 1e9))
