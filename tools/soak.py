"""Soak run: a long streaming sequence through HandTracker::update's unit of work, twice, compared bit for bit.

Every one of the B trackers follows the animation bank: in update k tracker i sees bench frame (i + k) mod 1024 (bench_data/frames1024.npz: rendered rows
3 + 9 j of the bank, so consecutive frames are 9 rows apart -- a fast hand -- and the sequence wraps: resets, CNN takes, the initializing countdown all occur).
Checked: run 1 == run 2 after every update (np.array_equal on the user poses: a second context, so allocation addresses differ), no NaN leaves the tracker,
device memory in use does not grow between update 10 and the last, the capacity counters (ht_capacity_events).  Prints one summary line per run + a JSON line.

    python tools/soak.py [updates=400] [B=1024]
"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from hand_tracking_samples_amd import native, weights as W

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
depth, cams, start = bench._load_frames(1024)
depth = depth.reshape(1024, 64, 64)


def run(tag):
    c = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B)
    c.load_weights(W.make_cnnb(W.DEFAULT_SEED, W.DEFAULT_FC2_GAIN)); c.set_params(microforce=3.0, mainthreadpasses=3)
    c.tracker_reset(start[np.arange(B) % 1024])
    sums = np.zeros(N, np.float64); nans = 0; used10 = None; resets = 0; t0 = time.time(); last = None
    for k in range(N):
        idx = (np.arange(B) + k) % 1024
        p = c.update_sync(depth[idx], cams[idx])
        nans += int(np.isnan(p).any(axis=(1, 2)).sum())
        sums[k] = float(np.nan_to_num(p.astype(np.float64)).sum())
        resets += int(c.debug_reset_flags(B).sum())
        if k == 10:
            torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); used10 = total - free
        last = p
        if (k + 1) % 100 == 0:
            print("  %s: %d updates, %.1f s" % (tag, k + 1, time.time() - t0), flush=True)
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info()
    cap = c.capacity_events()
    c.close()
    return {"sums": sums, "last": last, "nan_frames": nans, "mem_growth_bytes": int((total - free) - used10) if used10 is not None else None, "capacity_events": cap, "full_resets": resets, "seconds": round(time.time() - t0, 1)}


a = run("run 1")
b = run("run 2")
same = bool(np.array_equal(a["sums"], b["sums"]) and np.array_equal(a["last"], b["last"]))
out = {"updates": N, "frames_per_update": B, "frame_updates": N * B, "runs_equal_bit_for_bit": same,
       "nan_frames": [a["nan_frames"], b["nan_frames"]], "device_memory_growth_bytes_update10_to_end": [a["mem_growth_bytes"], b["mem_growth_bytes"]],
       "capacity_events_polytope_contacts_angular": [list(a["capacity_events"]), list(b["capacity_events"])], "full_resets": [a["full_resets"], b["full_resets"]], "seconds": [a["seconds"], b["seconds"]]}
print(json.dumps(out))
sys.exit(0 if same and a["nan_frames"] == 0 and b["nan_frames"] == 0 else 1)
