"""Per-frame k_solve cycle statistics on the bench workload (tuning aid).  Run with HT_DEBUG_SKIP=2048."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HT_DEBUG_SKIP", "2048")
from hand_tracking_samples_amd import native, weights  # noqa: E402

B = int(os.environ.get("FRAMES", "1024"))
CFG5 = os.environ.get("CONFIG5") == "1"      # BASELINE configs[4]: 128x128 frames, 26-bone hand, full-frame update
d = np.load(os.path.join(ROOT, "bench_data", "frames5_256.npz" if CFG5 else "frames1024.npz"))
idx = np.arange(B) % len(d["depth"])
depth, cams, start = d["depth"][idx], d["cam"][idx], d["startpose"][idx]
ctx = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand26.htfx" if CFG5 else "model_hand17.htfx"), B)
ctx.load_weights(weights.make_cnnb())
ctx.set_params(microforce=3.0, mainthreadpasses=int(os.environ.get('PASSES', '3')))
ctx.debug_solve_stats(B, reset=True); ctx.debug_contact_stats(B, reset=True)
for it in range(2):
    ctx.tracker_reset(start)
    if CFG5:
        ctx.update_frames_sync(depth, cams, 0.17)
    else:
        ctx.update_sync(depth, cams)
    st = ctx.debug_solve_stats(B, reset=True)
    cs = ctx.debug_contact_stats(B, reset=True)
n = st[:, 0:1]
print("launches/frame", np.unique(st[:, 0]))
names = ["chain", "linear", "angular", "sweeps_total"]
print("prologue parts, mean cycles/frame: state+rays %.0f, angular rows %.0f, linear groups %.0f, level schedule %.0f, chain lists+records %.0f" % (tuple(st[:, 12:16].mean(axis=0)) + ((st[:, 11] - st[:, 12:16].sum(axis=1)).mean(),)))
print("prologue (entry to first sweep) cycles/frame: mean %.0f p90 %.0f max %.0f" % (st[:, 11].mean(), np.percentile(st[:, 11], 90), st[:, 11].max()))
for k, nm in enumerate(names):
    c = st[:, 1 + k]
    print("%-12s cycles/frame: mean %.0f  p50 %.0f  p90 %.0f  max %.0f (frame %d)" % (nm, c.mean(), np.median(c), np.percentile(c, 90), c.max(), c.argmax()))
for k, nm in zip(range(5, 11), ["steps_lin", "steps_ang", "maxchain", "n1", "n2", "na"]):
    c = st[:, k] / st[:, 0]
    print("%-12s per launch: mean %.1f  p90 %.1f  max %.1f" % (nm, c.mean(), np.percentile(c, 90), c.max()))
w = st[:, 4].argmax()
print("worst frame", w, st[w])

hd = ctx.debug_solve_tables_header(B) if os.environ.get("HT_TABLES", "0") != "0" else None      # the tables exist only when a run asked for them
if hd is not None: print("---- k_solve_prep, the update's last launch (a main-thread pass), cycles per frame: angular wave %.0f, joints wave %.0f, chains wave (lists) %.0f, boundary-plane wave %.0f (max %d), all waves to the barrier %.0f, chain couplings %.0f; tables ok on %d of %d frames"
      % (hd[:, 20].mean(), hd[:, 21].mean(), hd[:, 22].mean(), hd[:, 23].mean(), hd[:, 23].max(), hd[:, 24].mean(), hd[:, 25].mean(), int((hd[:, 0] != 0).sum()), B))
print("---- k_contacts (sum over %s launches per frame)" % np.unique(cs[:, 0]))
# the fields of k_contacts_coop (wave f of a block reports its own cycles beside frame f's counts); the lane-per-pair kernel (HT_CONTACTS_LANES) fills them differently
for k, nm in zip(range(1, 12), ["owner_cycles", "epa_phase_cycles", "epa_runs", "iterations", "scan_cycles", "scan_and_barriers", "total_cycles", "candidates", "after_pass_cycles", "pass_cycles", "prologue_cycles"]):
    c = cs[:, k]
    print("%-13s per frame: mean %.0f  p50 %.0f  p90 %.0f  max %.0f (frame %d)" % (nm, c.mean(), np.median(c), np.percentile(c, 90), c.max(), c.argmax()))
order = np.argsort(-cs[:, 7])[:6]
print("slowest blocks (per launch): owner, epa_phase, epa_runs, iterations, scan, scan+barriers, total, candidates, prologue")
for f in order:
    r = cs[f] / cs[f, 0]
    print("  frame %4d: %7.0f %7.0f %5.2f %5.1f %7.0f %7.0f %7.0f %5.1f %7.0f" % (f, r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[11]))
for k, nm in zip(range(12, 17), ["epa_face", "epa_support", "epa_surgery", "epa_iters", "epa_jobs"]):
    c = cs[:, k]
    print("%-13s of the frame's wave: mean %.0f  p50 %.0f  p90 %.0f  max %.0f (frame %d)" % (nm, c.mean(), np.median(c), np.percentile(c, 90), c.max(), c.argmax()))
j = cs[:, 16].sum()
if j > 0:
    print("per polytope run: face %.0f support %.0f surgery %.0f cycles, %.1f iterations" % (cs[:, 12].sum() / j, cs[:, 13].sum() / j, cs[:, 14].sum() / j, cs[:, 15].sum() / j))
