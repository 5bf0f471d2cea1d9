"""Diagnostic: the device's per-frame deviation from the reference on the 1024 bench frames beside the reference's own FMA-build spread per frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import htfx, oracle_lib as ol
from hand_tracking_samples_amd import native, weights as W
FR = np.load(os.path.join(ROOT, "bench_data/frames1024.npz")); REF = htfx.load(os.path.join(ROOT, "tests/golden/poses1024.htfx")); SP = np.load(os.path.join(ROOT, "tests/golden/ref_spread1024.npz"))
N = 1024
ctx = native.Context(ol.MODEL, N); ctx.load_weights(W.make_cnnb()); ctx.set_params(microforce=3.0, mainthreadpasses=3)
ctx.tracker_reset(FR["startpose"]); got = ctx.update_sync(FR["depth"].reshape(N, -1), FR["cam"]); other = ctx.get_state(1, N)[:, :, :7]
ref = REF["uw_pose_user"]
dp = np.abs(got[:, :, :3] - ref[:, :, :3]).max(axis=(1, 2)); dq = np.minimum(np.abs(got[:, :, 3:] - ref[:, :, 3:]), np.abs(got[:, :, 3:] + ref[:, :, 3:])).max(axis=(1, 2))
sp = np.maximum(SP["fma_on_user_dpos"], SP["fma_fast_user_dpos"]); sq = np.maximum(SP["fma_on_user_dquat"], SP["fma_fast_user_dquat"])
print("device: tight %d loose %d; p50 %.2e p90 %.2e p99 %.2e max %.2e | quat p50 %.2e p99 %.2e max %.2e" % (((dp <= 2e-5) & (dq <= 2e-4)).sum(), ((dp <= 2e-4) & (dq <= 2e-3)).sum(), *np.percentile(dp, [50, 90, 99, 100]), *np.percentile(dq, [50, 99, 100])))
out = np.nonzero((dp > 2e-5) | (dq > 2e-4))[0]
for i in out:
    print("frame %4d: device %.2e / %.2e   reference FMA builds %.2e / %.2e  (rank of sensitivity %d of 1024)" % (i, dp[i], dq[i], sp[i], sq[i], (sq > sq[i]).sum()))
print("reference-sensitive frames (FMA builds outside tight):", np.nonzero((sp > 2e-5) | (sq > 2e-4))[0].tolist())
print("flags equal:", np.array_equal(ctx.tracker_flags(N)[1], REF["flags"][:, 1].astype(np.int32)))
do = np.abs(other - REF["other_pose"]).max(axis=(1, 2)); so = np.maximum(np.maximum(SP["fma_on_other_dpos"], SP["fma_fast_other_dpos"]), np.maximum(SP["fma_on_other_dquat"], SP["fma_fast_other_dquat"]))
print("othermodel: device p50 %.2e p90 %.2e p99 %.2e max %.2e; reference FMA builds p50 %.2e p90 %.2e p99 %.2e max %.2e" % (*np.percentile(do, [50, 90, 99, 100]), *np.percentile(so, [50, 90, 99, 100])))
ctx.close()
