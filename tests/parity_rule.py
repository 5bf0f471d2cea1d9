"""The acceptance rule for device poses against the reference's, frame by frame -- ONE definition, used by tests/test_gpu_batch_parity.py and by bench.py's `verify`
(test infrastructure: nothing here is used by the product path).

The device solver is one more floating-point BUILD of the reference's algorithm (the same rows in the same order, another association order: csrc/ht_quad.hpp,
csrc/ht_block.hpp; tests/test_gpu_exact_solver.py pins that as its only difference).  How far a frame moves under such a change is a property of the frame: a few sit next
to a discrete decision and amplify a rounding difference a thousandfold in every build.  tests/golden/ref_spread1024*.npz hold, per frame, how far the reference's own two
FMA-contracted builds move from its IEEE build (tests/golden/ref_flag_spread.py): that is the yardstick.

  * every value finite;
  * a frame is inside the TIGHT band (2e-5 m, 2e-4 on the quaternions) -- or the reference's own builds leave the tight band on that frame too, and then a component
    (position / quaternion) that is outside its band is so by at most TWICE what the reference's builds move it;
  * no frame beyond the absolute cap (5e-3 m, 5e-2; with always_take_cnn 1e-1 m, 1.0: a sanity bound), whatever the reference's builds do;
  * medians at rounding level.
The quaternion distance takes one sign per quaternion (q and -q are the same rotation), not per component.
"""
import numpy as np

TIGHT = (2e-5, 2e-4)
LOOSE = (2e-4, 2e-3)
CAP = (5e-3, 5e-2)
CAP_TAKE_CNN = (1e-1, 1.0)      # always_take_cnn: the user pose follows the hard-driven othermodel, which the reference's own FMA builds move by up to 2.3e-2 m / 0.36 on single frames


def pose_diff(got, ref):
    """per frame: the largest position difference over the bodies, the largest quaternion distance min(|q - r|, |q + r|) over the bodies"""
    dp = np.abs(got[:, :, :3] - ref[:, :, :3]).max(axis=(1, 2))
    a = np.abs(got[:, :, 3:7] - ref[:, :, 3:7]).max(axis=2)
    b = np.abs(got[:, :, 3:7] + ref[:, :, 3:7]).max(axis=2)
    return dp, np.minimum(a, b).max(axis=1)


def spread_of(npz, which, frames=None):
    """the frame's sensitivity: the larger move of the reference's two FMA builds; which = "user" | "other" """
    sp = np.maximum(npz["fma_on_%s_dpos" % which], npz["fma_fast_%s_dpos" % which])
    sq = np.maximum(npz["fma_on_%s_dquat" % which], npz["fma_fast_%s_dquat" % which])
    return (sp, sq) if frames is None else (sp[frames], sq[frames])


def frame_rule(dp, dq, sp, sq, factor=2.0, cap=CAP):
    """ok[i], tight[i] for every frame"""
    finite = np.isfinite(dp) & np.isfinite(dq)
    tight = finite & (dp <= TIGHT[0]) & (dq <= TIGHT[1])
    ref_tight = (sp <= TIGHT[0]) & (sq <= TIGHT[1])
    within = ((dp <= TIGHT[0]) | (dp <= factor * sp)) & ((dq <= TIGHT[1]) | (dq <= factor * sq))
    ok = finite & (tight | (~ref_tight & within)) & (dp <= cap[0]) & (dq <= cap[1])
    return ok, tight


def cross_build_failures(npz, which, factor=2.0, cap=CAP, frames=None):
    """The allow-list for poses that cannot be held frame by frame by NAME (CNN-driven poses, configs[4]'s model): hold one of the reference's own two FMA builds the way
    the device is held -- frame_rule against the OTHER build's move as the yardstick -- and count the frames that fail; the smaller of the two directions.  Two legitimate
    builds of the reference fail each other's per-frame rule on that many frames (the frames next to a discrete decision, which every perturbation moves differently);
    the device may fail on no more."""
    a = (npz["fma_on_%s_dpos" % which], npz["fma_on_%s_dquat" % which]); b = (npz["fma_fast_%s_dpos" % which], npz["fma_fast_%s_dquat" % which])
    if frames is not None:
        a = (a[0][frames], a[1][frames]); b = (b[0][frames], b[1][frames])
    ok_ab, _ = frame_rule(a[0], a[1], b[0], b[1], factor, cap); ok_ba, _ = frame_rule(b[0], b[1], a[0], a[1], factor, cap)
    return int(min((~ok_ab).sum(), (~ok_ba).sum()))


def summary(got, ref, sp, sq, factor=2.0, median_tol=(2e-6, 4e-5), cap=CAP):
    """dict for bench.py's verify block / the tests' assertions"""
    finite = bool(np.isfinite(got).all())
    dp, dq = pose_diff(np.nan_to_num(got, nan=1e9, posinf=1e9, neginf=1e9), ref)
    ok, tight = frame_rule(dp, dq, sp, sq, factor, cap)
    loose = (dp <= LOOSE[0]) & (dq <= LOOSE[1])
    med = bool(np.median(dp) <= median_tol[0] and np.median(dq) <= median_tol[1])
    bad = np.nonzero(~ok)[0]
    return {"ok": bool(finite and ok.all() and med), "finite": finite, "frames": int(len(dp)), "frames_failing_the_rule": [int(i) for i in bad[:16]],
            "within_2e-5m_2e-4": int(tight.sum()), "within_2e-4m_2e-3": int(loose.sum()),
            "reference_fma_builds_within_2e-5m_2e-4": int(((sp <= TIGHT[0]) & (sq <= TIGHT[1])).sum()),
            "median_abs_dpos_m": float(np.median(dp)), "median_abs_dquat": float(np.median(dq)), "max_abs_dpos_m": float(dp.max()), "max_abs_dquat": float(dq.max()),
            "worst_frame": int(np.argmax(dp / np.maximum(sp, TIGHT[0])))}


def distribution(got, ref, sp, sq, cap=CAP_TAKE_CNN, max_factor=None):
    """For poses that a rounding difference moves on frames of its own choosing (CNN-driven poses; configs[4]'s ill-conditioned 26-bone model): which frames amplify depends on
    the perturbation -- the reference's own two FMA builds disagree with each other there -- so the frames outside the band are held by number and size, not by name:
    percentiles p50 / p90 / p99 at most twice the reference's builds', no more frames outside the tight band than theirs, the largest inside the absolute cap and -- with
    max_factor -- at most max_factor times their largest.  Returns (ok, dict)."""
    finite = bool(np.isfinite(got).all())
    dp, dq = pose_diff(np.nan_to_num(got, nan=1e9, posinf=1e9, neginf=1e9), ref)
    pd, ps = np.percentile(np.maximum(dp, dq), [50, 90, 99]), np.percentile(np.maximum(sp, sq), [50, 90, 99])
    nd, ns = int(((dp > TIGHT[0]) | (dq > TIGHT[1])).sum()), int(((sp > TIGHT[0]) | (sq > TIGHT[1])).sum())
    mp, mq = (cap[0], cap[1]) if max_factor is None else (min(cap[0], max(max_factor * float(sp.max()), TIGHT[0])), min(cap[1], max(max_factor * float(sq.max()), TIGHT[1])))
    ok = bool(finite and (pd <= 2 * ps).all() and nd <= ns and dp.max() <= mp and dq.max() <= mq)
    return ok, {"p50_p90_p99": [float(x) for x in pd], "reference_fma_builds_p50_p90_p99": [float(x) for x in ps], "frames_outside_2e-5m_2e-4": nd, "reference_fma_builds_outside": ns,
                "max_abs_dpos_m": float(dp.max()), "max_abs_dquat": float(dq.max()), "reference_fma_builds_max": [float(sp.max()), float(sq.max())],
                "within_2e-5m_2e-4": int(len(dp) - nd), "within_2e-4m_2e-3": int(((dp <= LOOSE[0]) & (dq <= LOOSE[1])).sum())}
