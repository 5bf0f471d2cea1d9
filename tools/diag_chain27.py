"""Diagnostic: the 27-body chain of tests/test_gpu_edges.py::test_models_with_many_ranged_joints, product build against the restatement, body by body."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import test_gpu_edges as tg  # noqa: E402
from hand_tracking_samples_amd import native, weights as wmod  # noqa: E402

w = wmod.make_cnnb()
tmp = tempfile.mkdtemp()
ok_json, baked = os.path.join(tmp, "chain27.json"), os.path.join(tmp, "chain27.htfx")
tg._chain_model_json(ok_json, 27)
native.model_bake(ok_json, baked)
depth, cams, _ = tg._bank(2)
start = np.zeros((2, 27, 7), np.float32); start[:, :, 6] = 1.0
for b in range(27):
    start[:, b, :3] = (-0.05 + 0.004 * b, 0.01 * np.sin(0.4 * b), 0.45 + 0.002 * b)
L = ol.lib()
steps = int(os.environ.get("STEPS", "1"))
res = {}
for build in (0, 5):
    ctx = native.Context(ok_json, 2)
    ctx.load_weights(w)
    ctx.set_params(microforce=3.0, mainthreadpasses=1, steps=steps)
    if build:
        ctx.debug_solver_build(build)
    ctx.tracker_reset(start)
    got, cnn = ctx.update_sync(depth, cams, want_cnn=True)
    other = ctx.get_state(1, 2)
    pfe, ini = ctx.tracker_flags(2)
    print("build", build, "capacity", ctx.capacity_events(), "prev_frame_error", pfe, "initializing", ini)
    ctx.close()
    res[build] = (got, other)
for k in range(2):
    a, e = res[0][1][k], res[5][1][k]
    dp = np.abs(a[:, :3] - e[:, :3]).max(axis=1)
    dq = np.minimum(np.abs(a[:, 3:7] - e[:, 3:7]), np.abs(a[:, 3:7] + e[:, 3:7])).max(axis=1)
    dqs = np.abs(a[:, 3:7] - e[:, 3:7]).max(axis=1)
    print("frame", k, "othermodel product vs exact: |dpos| max %.2e |dquat| (up to sign) max %.2e, signed max %.2e" % (dp.max(), dq.max(), dqs.max()))
    for b in range(27):
        if dqs[b] > 1e-3 or dp[b] > 1e-4:
            print("   body", b, "dpos %.2e dquat %.2e signed %.2e" % (dp[b], dq[b], dqs[b]), a[b, 3:7], e[b, 3:7])

# the reset branch alone (PoseFromScratch + k rounds of UnibodyFit), product against exact-order build, round by round
ctx = native.Context(ok_json, 2)
ctx.load_weights(w); ctx.set_params(microforce=3.0, mainthreadpasses=1, steps=steps)
ctx.tracker_reset(start); ctx.update_sync(depth, cams)
an = ctx.cnn_results(2)[2]
ctx.close()
for k in range(4):
    st = {}
    for build in (0, 5):
        ctx = native.Context(ok_json, 2)
        ctx.load_weights(w); ctx.set_params(microforce=3.0, mainthreadpasses=1, steps=steps)
        if build:
            ctx.debug_solver_build(build)
        ctx.stage_prepare(depth, cams); ctx.tracker_reset(start)
        ctx.stage_scratch_unibody(an, 2, k)
        st[build] = ctx.get_state(1, 2)
        ctx.close()
    for f in range(2):
        a, e = st[0][f], st[5][f]
        print("reset branch, %d unibody rounds, frame %d: |dpos| %.2e |dquat| %.2e" % (k, f, np.abs(a[:, :3] - e[:, :3]).max(), np.minimum(np.abs(a[:, 3:7] - e[:, 3:7]), np.abs(a[:, 3:7] + e[:, 3:7])).max()))
