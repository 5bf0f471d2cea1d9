"""Trains the hand-pose net with the repo's own training step (ht_cnn_train = CNN::Train, cnn.h:558-580) the way train-hand-pose-cnn does (train-cnn.cpp:156-162: one sample
per step, lr 0.001, labels from GatherHandExpectedCNN of the frame's ground-truth pose, handtrack.h:160-173) on the bench's software-rendered 64x64 tiles
(bench_data/frames1024.npz: animation-bank rows 3 + 9 i with their ground-truth poses), from seeded Xavier weights with FC2 gain 1 (the reference's own init range,
cnn.h:282,448).  The trained handposedd.cnnb is not shipped with the reference: this is how the repo makes a net whose heat-maps have REAL peaks.

    python tools/train_synthetic.py [--epochs 300] [--curve profiles/r06_train_curve.json] [--out weights.cnnb]

Every 16th frame is held out.  Deterministic: seeded weights, seeded shuffles, batch-1 SGD on one stream.  The weights are not committed -- seed + script are."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand_tracking_samples_amd import native, weights as W  # noqa: E402

SEED = 0x5EED0001
HOLD = 16      # every HOLD-th frame is held out


def dataset():
    d = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
    n = len(d["gtpose"])
    return d["depth"][:n].reshape(n, -1), d["cam"][:n], d["gtpose"]


def inputs_and_labels(ctx, depth, cams, gt):
    """cnn_input of every tile (handtrack.h:700, on the device) and its expected output (GatherHandExpectedCNN on the host library)"""
    n = len(depth)
    x = np.zeros((n, 4096), np.float32)
    B = ctx.max_batch
    for i in range(0, n, B):
        m = min(B, n - i)
        x[i:i + m] = ctx.stage_prepare(depth[i:i + m], cams[i:i + m])[0]
    t = np.stack([native.expected_cnn(gt[i], cams[i]) for i in range(n)])
    return x, t


def train(epochs, log=None, model=None, batch=256):
    depth, cams, gt = dataset()
    n = len(depth)
    ctx = native.Context(model or os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), batch)
    ctx.load_weights(W.make_cnnb(SEED, 1.0))
    x, t = inputs_and_labels(ctx, depth, cams, gt)
    test = np.arange(0, n, HOLD); tr = np.setdiff1d(np.arange(n), test)
    rng = np.random.default_rng(SEED)

    def held_out():
        y = np.concatenate([ctx.cnn_eval(x[test[i:i + batch]]) for i in range(0, len(test), batch)])
        return float(((y - t[test]) ** 2).mean()), y

    curve = []
    mse0, _ = held_out()
    t0 = time.perf_counter()
    for ep in range(epochs):
        perm = rng.permutation(tr)
        mse = ctx.cnn_train(x[perm], t[perm], 0.001)
        rec = {"epoch": ep + 1, "steps": (ep + 1) * len(tr), "train_mse": float(mse.mean())}
        if (ep + 1) % 10 == 0 or ep + 1 == epochs:
            rec["held_out_mse"], _ = held_out()
        curve.append(rec)
        if log and ((ep + 1) % 10 == 0 or ep == 0):
            log("epoch %4d: train mse %.3e%s (%.0f s)" % (ep + 1, rec["train_mse"], (", held out %.3e" % rec["held_out_mse"]) if "held_out_mse" in rec else "", time.perf_counter() - t0))
    w = ctx.cnn_get_weights()
    _, y = held_out()
    ctx.close()
    return {"weights": w, "curve": curve, "held_out_mse_before": mse0, "test_frames": test, "test_outputs": y, "inputs": x, "labels": t, "seconds": time.perf_counter() - t0, "train_frames": int(len(tr))}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--curve", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = train(a.epochs, log=lambda s: print(s, flush=True))
    c = r["curve"]
    peak = r["test_outputs"][:, :2048].reshape(-1, 8, 256).max(axis=2)
    print("held-out mse %.3e -> %.3e (x %.1f); train mse first epoch %.3e, last %.3e; mean landmark peak of the held-out heat-maps %.3f (uniform: 0.004); %d steps in %.0f s"
          % (r["held_out_mse_before"], c[-1]["held_out_mse"], r["held_out_mse_before"] / c[-1]["held_out_mse"], c[0]["train_mse"], c[-1]["train_mse"], float(peak.mean()), c[-1]["steps"], r["seconds"]))
    if a.curve:
        json.dump({"what": "tools/train_synthetic.py: batch-1 SGD (ht_cnn_train), lr 0.001, seeded Xavier init (gain 1), %d train / %d held-out tiles of bench_data/frames1024.npz" % (r["train_frames"], len(r["test_frames"])),
                   "held_out_mse_before": r["held_out_mse_before"], "curve": c, "seconds": r["seconds"]}, open(a.curve, "w"), indent=1)
    if a.out:
        W.save_cnnb(a.out, r["weights"])
