"""The algebra behind k_reset's single-body solve sixteen rows at a time (csrc/ht_quad.hpp: quad_block16_step), checked in float64 on random rows.

LimitLinear::Iter (/root/reference/third_party/physics.h:289-307) on a row j of ONE body, in the Jacobian form the kernels use:
    x_j = -ts_j - c_j . M           (c_j = (n*massinv, Iinv*g) / effective mass, M = the body's momenta (P, L))
    imp_j = clamp(x_j, lo_j - sum_j, hi_j - sum_j);   M += d_j * imp_j   (d_j = (n, g));   sum_j += imp_j
Rows are applied strictly in order (Gauss-Seidel).  Within a block of sixteen rows the momenta seen by row j are M0 + sum_{i<j} d_i imp_i, so
    x_j = (-ts_j - c_j . M0) - sum_{i<j} G[j, i] * imp_i,      G[j, i] = c_j . d_i
which lets the sixteen c_j . M0 be formed side by side and leaves a short serial resolve.  Same rows, same order, same clamps: in exact arithmetic the two
are the same sweep; in float32 they differ by rounding (tests/test_gpu_batch_parity.py holds the device to the reference on that)."""
import numpy as np


def _rows(rng, n):
    d = rng.normal(size=(n, 6))                                             # (n, g) of every row
    minv = np.concatenate([np.full(3, 2.5), rng.uniform(50.0, 400.0, 3)])   # inverse mass and a (diagonal) inverse inertia: positive definite, as a body's is
    c = d * minv / ((d * minv) * d).sum(axis=1, keepdims=True)              # (n*massinv, Iinv*g) / effective mass: the sweeps of such rows converge
    ts = rng.normal(size=n) * 0.1
    lo = -np.abs(rng.normal(size=n)) * 0.05
    hi = np.abs(rng.normal(size=n)) * 0.05
    lo[::7] = 0.0      # one-sided rows (contacts, chamber planes)
    return c, d, ts, lo, hi


def _sweep_row_by_row(M, sums, c, d, ts, lo, hi):
    for j in range(len(ts)):
        x = -ts[j] - c[j] @ M
        imp = min(max(x, lo[j] - sums[j]), hi[j] - sums[j])
        M = M + d[j] * imp
        sums[j] += imp
    return M


def _sweep_blocked(M, sums, c, d, ts, lo, hi, W=16):
    n = len(ts)
    for b0 in range(0, n, W):
        j1 = min(n, b0 + W)
        cb, db = c[b0:j1], d[b0:j1]
        G = cb @ db.T                                   # G[j, i] = c_j . d_i
        x = -ts[b0:j1] - cb @ M                         # all rows of the block against the momenta before it
        imp = np.zeros(j1 - b0)
        for i in range(j1 - b0):                        # resolved in row order
            imp[i] = min(max(x[i], lo[b0 + i] - sums[b0 + i]), hi[b0 + i] - sums[b0 + i])
            x[i + 1:] -= G[i + 1:, i] * imp[i]
        M = M + db.T @ imp                              # every row's contribution, summed in any order
        sums[b0:j1] += imp
    return M


def test_blocked_sweeps_equal_row_by_row_sweeps():
    rng = np.random.default_rng(11)
    for n in (1, 15, 16, 17, 150, 448):
        c, d, ts, lo, hi = _rows(rng, n)
        Ma = rng.normal(size=6) * 0.01; Mb = Ma.copy()
        sa = np.zeros(n); sb = np.zeros(n)
        for sweep in range(20):
            Ma = _sweep_row_by_row(Ma, sa, c, d, ts, lo, hi)
            Mb = _sweep_blocked(Mb, sb, c, d, ts, lo, hi)
        assert np.abs(Ma - Mb).max() <= 1e-11 * max(1.0, np.abs(Ma).max()), n
        assert np.abs(sa - sb).max() <= 1e-11, n
        assert (np.abs(sa - lo) < 1e-12).any() or (np.abs(sa - hi) < 1e-12).any() or n < 15      # the clamps are active on some rows (the comparison covers them)


def test_rows_that_change_nothing_fill_a_block():
    """A block is filled up with zero rows (zero direction, zero limits): they take no impulse and pass none on."""
    rng = np.random.default_rng(12)
    c, d, ts, lo, hi = _rows(rng, 21)
    pad = 32 - 21
    cp = np.vstack([c, np.zeros((pad, 6))]); dp = np.vstack([d, np.zeros((pad, 6))])
    tsp = np.concatenate([ts, np.zeros(pad)]); lop = np.concatenate([lo, np.zeros(pad)]); hip = np.concatenate([hi, np.zeros(pad)])
    M0 = rng.normal(size=6) * 0.01
    sa = np.zeros(21); sb = np.zeros(32)
    Ma = _sweep_blocked(M0.copy(), sa, c, d, ts, lo, hi)
    Mb = _sweep_blocked(M0.copy(), sb, cp, dp, tsp, lop, hip)
    assert np.array_equal(Ma, Mb) and np.array_equal(sa, sb[:21]) and not sb[21:].any()


def test_blocks_of_four_rows_equal_row_by_row_sweeps():
    """csrc/ht_quad.hpp: quad_block_step (round 5: k_solve's single-body rows four at a time on the four quads of a DPP row; a chain padded to whole blocks with rows that
    change nothing): the same algebra with W = 4."""
    rng = np.random.default_rng(12)
    for n in (1, 3, 4, 5, 109, 284):
        c, d, ts, lo, hi = _rows(rng, n)
        pad = (-n) % 4      # the record that changes nothing: zero direction, zero limits
        cp, dp_, tsp, lop, hip = np.vstack([c, np.zeros((pad, 6))]), np.vstack([d, np.zeros((pad, 6))]), np.concatenate([ts, np.zeros(pad)]), np.concatenate([lo, np.zeros(pad)]), np.concatenate([hi, np.zeros(pad)])
        Ma = rng.normal(size=6) * 0.01; Mb = Ma.copy()
        sa = np.zeros(n); sb = np.zeros(n + pad)
        for sweep in range(20):
            Ma = _sweep_row_by_row(Ma, sa, c, d, ts, lo, hi)
            Mb = _sweep_blocked(Mb, sb, cp, dp_, tsp, lop, hip, W=4)
        assert np.abs(Ma - Mb).max() <= 1e-11 * max(1.0, np.abs(Ma).max()), n
        assert np.abs(sa - sb[:n]).max() <= 1e-11 and np.all(sb[n:] == 0.0), n
