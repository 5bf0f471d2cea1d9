"""The algebra behind k_solve's two-body rows resolved a block at a time (csrc/ht_block.hpp), checked in float64 on random articulated rows.

LimitLinear::Iter / LimitAngular::Iter (/root/reference/third_party/physics.h:289-307, 251-265) on rows that couple TWO bodies, in the Jacobian form the kernels use:
    x_j = (T_j - c_j . M) k_j        (M = the momenta (P, L) of all bodies, c_j = the row's velocity coefficients on its two bodies, k_j = 1 / effective mass)
    d_j = clamp(x_j, lo_j - sum_j, hi_j - sum_j);   M += D_j d_j   (D_j = what a unit impulse adds to its two bodies);   sum_j += d_j
strictly in row order (Gauss-Seidel), friction rows limited by their normal row's CURRENT sum (physics.h:292).  Within a block of consecutive rows the momenta seen by
row j are M0 + sum_{i<j} D_i d_i, so
    x_j = (T_j - c_j . M0) k_j + sum_{i<j} G[j, i] d_i,      G[j, i] = -k_j c_j . D_i      (zero unless rows i and j share a body)
which lets a block's c_j . M0 be formed side by side (one row per lane) and leaves the serial resolve clamp / broadcast / multiply-add.  The momenta are brought up to
date once per block by summing D_i d_i per body over the block's (row, side) edges sorted by body -- any fixed order.  Same rows, same order, same clamps: in exact
arithmetic the same sweep; in float32 they differ by rounding (tests/test_gpu_batch_parity.py holds the device to the reference on that)."""
import numpy as np


def _scene(rng, nb, nrows, ncontact_triples):
    """random two-body rows on a star-and-chains body graph like a hand's (body 1 = palm), the last 3 * ncontact_triples rows = contact triples (normal, 2 friction)"""
    minv = np.concatenate([np.full((nb, 3), 2.5), rng.uniform(50.0, 400.0, (nb, 3))], axis=1)      # per body: inverse mass (x3), a diagonal inverse inertia
    pairs = [(1, b) if b % 3 == 2 else (b - 1, b) for b in range(2, nb)] + [(0, 1)]
    rows = []
    for r in range(nrows):
        a, b = pairs[rng.integers(len(pairs))] if r % 5 else (-1, int(rng.integers(nb)))            # every fifth row has only one body (drive rows, cones against the world)
        Da, Db = rng.normal(size=6), rng.normal(size=6)
        if a < 0: Da[:] = 0
        rows.append((a, b, Da, Db))
    c = np.zeros((nrows, nb, 6)); D = np.zeros((nrows, nb, 6)); k = np.zeros(nrows)
    for r, (a, b, Da, Db) in enumerate(rows):
        if a >= 0: D[r, a] = Da; c[r, a] = Da * minv[a]
        D[r, b] = Db; c[r, b] = Db * minv[b]
        k[r] = 1.0 / (c[r] * D[r]).sum()                                                            # effective mass: the sweeps of such rows converge
    T = rng.normal(size=nrows) * 0.1
    lo = -np.abs(rng.normal(size=nrows)) * 0.05; hi = np.abs(rng.normal(size=nrows)) * 0.05
    lo[::7] = 0.0
    master = np.full(nrows, -1); mu = np.zeros(nrows)
    for t in range(ncontact_triples):
        n0 = nrows - 3 * (ncontact_triples - t)
        lo[n0] = 0.0; hi[n0] = 1e9
        for f in (1, 2): master[n0 + f] = n0; mu[n0 + f] = 0.6
    return c.reshape(nrows, -1), D.reshape(nrows, -1), k, T, lo, hi, master, mu


def _limits(j, sums, lo, hi, master, mu):
    if master[j] >= 0:
        lim = mu[j] * sums[master[j]]
        return -lim - sums[j], lim - sums[j]
    return lo[j] - sums[j], hi[j] - sums[j]


def _sweep_row_by_row(M, sums, c, D, k, T, lo, hi, master, mu):
    for j in range(len(T)):
        x = (T[j] - c[j] @ M) * k[j]
        l, h = _limits(j, sums, lo, hi, master, mu)
        d = min(max(x, l), h)
        M = M + D[j] * d
        sums[j] += d
    return M


def _sweep_blocked(M, sums, c, D, k, T, lo, hi, master, mu, W, rng):
    n = len(T)
    for b0 in range(0, n, W):
        j1 = min(n, b0 + W)
        cb, Db = c[b0:j1], D[b0:j1]
        G = -(k[b0:j1, None] * (cb @ Db.T))                     # G[j, i] = -k_j c_j . D_i
        x = (T[b0:j1] - cb @ M) * k[b0:j1]                       # every row against the momenta before the block
        d = np.zeros(j1 - b0)
        s0 = sums.copy()                                         # the sums before the block (a friction row's limits take its master's sum as of its own step)
        for i in range(j1 - b0):
            j = b0 + i
            if master[j] >= 0:
                ms = s0[master[j]] + (d[master[j] - b0] if master[j] >= b0 else 0.0)
                l, h = -mu[j] * ms - s0[j], mu[j] * ms - s0[j]
            else:
                l, h = lo[j] - s0[j], hi[j] - s0[j]
            d[i] = min(max(x[i], l), h)
            x[i + 1:] += G[i + 1:, i] * d[i]
        order = rng.permutation(j1 - b0)                         # the contributions per body, summed in any order
        for i in order:
            M = M + Db[i] * d[i]
        sums[b0:j1] += d
    return M


def test_blocked_two_body_sweeps_equal_row_by_row_sweeps():
    rng = np.random.default_rng(5)
    for nb, nrows, ntri, W in ((17, 48 + 15, 5, 30), (17, 84, 0, 32), (26, 120, 8, 30), (3, 9, 1, 30), (17, 31, 0, 32)):
        c, D, k, T, lo, hi, master, mu = _scene(rng, nb, nrows, ntri)
        Ma = rng.normal(size=nb * 6) * 0.01; Mb = Ma.copy()
        sa = np.zeros(nrows); sb = np.zeros(nrows)
        for sweep in range(20):
            Ma = _sweep_row_by_row(Ma, sa, c, D, k, T, lo, hi, master, mu)
            Mb = _sweep_blocked(Mb, sb, c, D, k, T, lo, hi, master, mu, W, rng)
        assert np.abs(Ma - Mb).max() <= 1e-10 * max(1.0, np.abs(Ma).max()), (nb, nrows)
        assert np.abs(sa - sb).max() <= 1e-10, (nb, nrows)
        assert ((np.abs(sa - lo) < 1e-12) | (np.abs(sa - hi) < 1e-12)).any() or nrows < 12      # clamps are active on some rows: the comparison covers them
        if ntri:
            f = np.nonzero(master >= 0)[0]
            assert (np.abs(np.abs(sa[f]) - mu[f] * sa[master[f]]) < 1e-9).any()                  # and some friction rows sit on their cone


def test_couplings_vanish_between_rows_without_a_common_body():
    rng = np.random.default_rng(6)
    c, D, k, T, lo, hi, master, mu = _scene(rng, 17, 40, 0)
    G = -(k[:, None] * (c @ D.T))
    cb = (np.abs(c.reshape(40, 17, 6)).sum(axis=2) > 0); Db = (np.abs(D.reshape(40, 17, 6)).sum(axis=2) > 0)
    share = (cb[:, None, :] & Db[None, :, :]).any(axis=2)
    assert (G[~share] == 0).all() and (G[share] != 0).any()


def test_the_fold_of_two_blocks_into_one_register_file():
    """ht_block.hpp's storage: lane m keeps G_A[m, i] of its forward row in register i < m and G_B[31 - m, 31 - r] of its backward row in register r > m; while block A is
    resolved lane m stays enabled through step m only, so register r >= m (block B's) is never applied to it -- and the other way round."""
    W = 32
    rng = np.random.default_rng(7)
    GA = np.tril(rng.normal(size=(W, W)), -1); GB = np.tril(rng.normal(size=(W, W)), -1)
    reg = np.zeros((W, W))                       # reg[lane, register]
    for m in range(W):
        for r in range(W):
            if r < m: reg[m, r] = GA[m, r]
            if r > m: reg[m, r] = GB[W - 1 - m, W - 1 - r]
    xa = rng.normal(size=W); xb = rng.normal(size=W)
    # block A forwards: step i resolves lane i, lanes > i stay enabled
    x = xa.copy(); da = np.zeros(W)
    for i in range(W):
        da[i] = np.clip(x[i], -0.3, 0.3)
        x[i + 1:] += reg[i + 1:, i] * da[i]
    # block B backwards: step p resolves lane 31 - p with register 31 - p, lanes < 31 - p stay enabled
    y = xb[::-1].copy(); db = np.zeros(W)        # y[lane]: row p sits on lane 31 - p
    for p in range(W):
        ln = W - 1 - p
        db[p] = np.clip(y[ln], -0.3, 0.3)
        y[:ln] += reg[:ln, W - 1 - p] * db[p]
    # against the plain triangular resolves
    x = xa.copy(); ra = np.zeros(W)
    for i in range(W):
        ra[i] = np.clip(x[i], -0.3, 0.3); x[i + 1:] += GA[i + 1:, i] * ra[i]
    x = xb.copy(); rb = np.zeros(W)
    for i in range(W):
        rb[i] = np.clip(x[i], -0.3, 0.3); x[i + 1:] += GB[i + 1:, i] * rb[i]
    assert np.array_equal(da, ra) and np.array_equal(db, rb)


def test_segmented_sums_over_edges_sorted_by_body():
    """the DPP scan of ht_block.hpp (row_shr 1, 2, 4, 8, row_bcast15, row_bcast31 with per-lane participation bits) in numpy: the last lane of every body's run ends up
    with the body's total"""
    rng = np.random.default_rng(8)
    for trial in range(50):
        nbod = rng.integers(1, 20)
        body = np.sort(rng.integers(0, nbod, 64)); nvalid = rng.integers(1, 65); valid = np.arange(64) < nvalid
        v = rng.normal(size=64) * valid
        same = lambda a, b: 0 <= b < 64 and valid[a] and valid[b] and body[a] == body[b]
        x = v.copy()
        for d in (1, 2, 4, 8):
            src = np.array([x[l - d] if (l - d >= (l & ~15)) else 0.0 for l in range(64)])
            m = np.array([1.0 if (l - d >= (l & ~15)) and same(l, l - d) else 0.0 for l in range(64)])
            x = x + m * src
        src = np.array([x[(l & ~15) - 1] if (l & 16) else 0.0 for l in range(64)])
        m = np.array([1.0 if (l & 16) and same(l, (l & ~15) - 1) else 0.0 for l in range(64)])
        x = x + m * src
        src = np.array([x[31] if l >= 32 else 0.0 for l in range(64)])
        m = np.array([1.0 if l >= 32 and same(l, 31) else 0.0 for l in range(64)])
        x = x + m * src
        for l in range(64):
            if valid[l] and not same(l, l + 1):
                assert abs(x[l] - v[valid & (body == body[l])].sum()) < 1e-12
