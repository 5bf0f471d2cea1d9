"""Latency of the drop-in surface: HandTracker::update, one frame per call with host buffers, as synthetic-tracker.cpp:215 calls it, and ht_update_sync on 8 / 64 trackers.
Builds tests/cxx_headless_driver.cpp and runs its `latency` mode on the first 32 bench frames (64x64 tiles).  Prints one JSON line per batch size."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hand_tracking_samples_amd import native, weights as W
native.load()
lib = os.path.dirname(native.lib_path())
with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "driver")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx_headless_driver.cpp"), "-o", exe, "-L" + lib, "-lht_mi355x", "-Wl,-rpath," + lib])
    z = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz")); n = 32
    with open(os.path.join(td, "in.bin"), "wb") as f:
        f.write(struct.pack("<4i", n, 64, 64, 17))
        for k in range(n):
            f.write(z["depth"][k].astype(np.uint16).tobytes()); f.write(z["cam"][k].astype(np.float32).tobytes()); f.write(z["startpose"][k].astype(np.float32).tobytes()); f.write(z["startpose"][k].astype(np.float32).tobytes())
    W.save_cnnb(os.path.join(td, "w.cnnb"), W.make_cnnb())
    iters = sys.argv[1] if len(sys.argv) > 1 else "300"
    sys.stdout.write(subprocess.check_output([exe, "latency", os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), os.path.join(td, "w.cnnb"), os.path.join(td, "in.bin"), iters]).decode())
