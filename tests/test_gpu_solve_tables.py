"""Round 6: a solve's tables -- the joints' linear groups, every angular record, the couplings and sorted edges of the two-body blocks, the per-body chain lists of the
single-body rows and their blocks' couplings -- are made by k_solve_prep (csrc/ht_prep.hip: four waves per frame, beside the contact kernel) instead of k_solve's one-wave
prologue.  ht_debug_solve_tables(0) puts them back where they were.  WHERE they are made must not show in a single bit: physmodel.h:345-351 (row order),
physics.h:556-562 (sweep order) are untouched, and every expression is the prologue's own."""
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _update(weights, tables, n, take_cnn=0, build=0, voxel=0):
    from hand_tracking_samples_amd import native
    d = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))
    depth, cams, start = d["depth"].reshape(1024, -1)[:n], d["cam"][:n], d["startpose"][:n]
    ctx = native.Context(ol.MODEL, n)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3, always_take_cnn=take_cnn, subsample_voxel=voxel)
        ctx.debug_solve_tables(tables)
        if build:
            ctx.debug_solver_build(build)
        ctx.tracker_reset(start)
        out = []
        for _ in range(2):      # the second update runs on carried momenta and tracker flags
            poses, _ = ctx.update_sync(depth, cams, want_cnn=True)
            out += [poses, ctx.get_state(0, n), ctx.get_state(1, n)]
        out += list(ctx.tracker_flags(n))
        assert ctx.capacity_events() == (0, 0, 0)
    finally:
        ctx.close()
    return out


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("take_cnn,build", [(0, 0), (1, 0), (0, 1), (0, 3), (0, 4), (0, 6)])
def test_update_is_the_same_bits_with_the_tables_made_beside_the_contacts(weights, take_cnn, build, mode):
    """256 bench frames (chamber on / off at 400 points, reset frames, up to a dozen contacts), every build of k_solve: LDS builds, the build whose arrays all live in HBM (4),
    the build with four angular-row slots (6)"""
    n = 256
    a, b = _update(weights, mode, n, take_cnn, build), _update(weights, 0, n, take_cnn, build)      # mode 1: every table from k_solve_prep; 2: the pose-only tables alone (the chain lists stay in k_solve)
    for k, (x, y) in enumerate(zip(a, b)):
        differ = np.nonzero((x != y).reshape(n, -1).any(axis=1))[0]
        assert len(differ) == 0, "output %d differs on frames %s (largest move %.3e)" % (k, differ[:8].tolist(), float(np.abs(x.astype(np.float64) - y).max()))


def test_stage_calls_are_the_same_bits(golden, weights):
    """MultiStepSim and one main-thread pass as stage calls on the golden frames (the stage entry points launch k_solve_prep themselves; the pass makes its own boundary planes)"""
    from hand_tracking_samples_amd import native
    nf = 8
    depth = np.stack([golden["f%d/depth" % f].reshape(-1) for f in range(nf)]); cams = np.stack([golden["f%d/cam" % f] for f in range(nf)])
    start = np.stack([golden["f%d/startpose" % f] for f in range(nf)])
    res = []
    for tables in (1, 2, 0):
        ctx = native.Context(ol.MODEL, nf)
        try:
            ctx.load_weights(weights)
            ctx.set_params(microforce=3.0, mainthreadpasses=3)
            ctx.debug_solve_tables(tables)
            ctx.tracker_reset(start)
            ctx.stage_prepare(depth, cams)
            an = ctx.stage_decode(np.stack([golden["f%d/cnn_output" % f].reshape(-1) for f in range(nf)]), cams)
            ctx.stage_multistep(an, nf)
            s1 = ctx.get_state(1, nf)
            ctx.stage_fit(nf); ctx.stage_fit(nf)
            res.append((s1, ctx.get_state(0, nf)))
        finally:
            ctx.close()
    for r in res[:2]:
        assert np.array_equal(r[0], res[2][0]) and np.array_equal(r[1], res[2][1])


def test_configs4_end_to_end_is_the_same_bits():
    """the 26-bone hand (25 joints: three linear blocks of joint rows, ~120 angular rows, more than sixteen bodies' chains) on 128x128 frames end to end, tables on == off"""
    from hand_tracking_samples_amd import native, weights as W
    fr = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames5_256.npz"))
    n = 128
    res = []
    for tables in (1, 2, 0):
        ctx = native.Context(os.path.join(HERE, "golden", "model_hand26.htfx"), n)
        try:
            ctx.load_weights128(W.make_cnnb128())
            ctx.set_params(microforce=3.0, mainthreadpasses=3)
            ctx.debug_solve_tables(tables)
            ctx.tracker_reset(fr["startpose"][:n])
            out = []
            for _ in range(2):
                poses, _ = ctx.update_direct_sync(fr["depth"][:n], fr["cam"][:n], 128, want_cnn=True)
                out += [poses, ctx.get_state(0, n), ctx.get_state(1, n)]
            if tables == 1:
                hd = ctx.debug_solve_tables_header(n)
                print("configs[4]: tables usable on %d of %d frames in the update's last solve (angular rows %d .. %d)" % (int((hd[:, 0] != 0).sum()), n, hd[:, 1].min(), hd[:, 1].max()))
            assert ctx.capacity_events() == (0, 0, 0)
            res.append(out)
        finally:
            ctx.close()
    for r in res[:2]:
        for k, (x, y) in enumerate(zip(r, res[2])):
            assert np.array_equal(x, y), "output %d differs (largest move %.3e)" % (k, float(np.abs(x.astype(np.float64) - y).max()))
