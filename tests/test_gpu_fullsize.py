"""Parity at BASELINE.json's full sizes (500,000 / 1,000,000 bodies at theta 0.5 and 0.3; the 8,000,000
bodies of configs[3] as ONE context) through size-independent properties, plus oracle checks:
  * keys sorted, sort is a permutation, gather consistent;
  * the tree is a tree: children partition their parent's body range, every body in exactly one
    leaf, every emitted cell branches, cells <= n-1, records <= 2n (+ padding), edges halve, child
    blocks start on 64-byte boundaries;
  * root mass/COM = total mass / mass-weighted mean (fp64 on the host);
  * strict kernel == oracle walking the GPU's tree, bit for bit, V/O/P counters too: over ALL bodies at
    500k / 1M (both thetas), on a 4096-body slab at 8M;
  * the FAST (benchmarked) kernel vs the ORACLE over the same bodies (all of them up to 1M), stated
    distribution; at 8M additionally fast vs strict over all bodies;
  * K = 10 whole steps (bh_step, fast kernel) vs oracle.Oracle.step at 500k, 1M theta 0.5 and 0.3, K = 3 at
    8M: stated distribution on positions and velocities (north_star: "after N steps");
  * a whole step conserves the body set and the sticky flags stay clear.
"""
import numpy as np
import pytest

from helpers import oparams

pytestmark = pytest.mark.gpu

CONFIGS = [(500_000, 0.5), (1_000_000, 0.5), (1_000_000, 0.3), (8_000_000, 0.5)]


def _tree_invariants_vectorised(rec, n):
    kind, first, count, s = rec["kind"], rec["first"], rec["count"], rec["s"]
    internal = np.flatnonzero(kind == 1)
    pad = kind == 3
    assert len(internal) <= n - 1 and len(rec) - int(pad.sum()) <= 2 * n and len(rec) <= 3 * n + 2
    assert count[internal].min() >= 2 and count[internal].max() <= 8   # every emitted cell branches
    starts = first[internal]
    assert np.all(starts % 2 == 0) and starts.min() >= 2                 # blocks on 64-byte boundaries
    odd = internal[count[internal] % 2 == 1]
    assert np.all(kind[first[odd] + count[odd]] == 3)                     # odd block -> one padding entry
    # every record except the root and the padding is the child of exactly one cell
    ref = np.zeros(len(rec), np.int32)
    for k in range(8):
        sel = count[internal] > k
        np.add.at(ref, starts[sel] + k, 1)
    assert ref[0] == 0 and pad[1] and np.all(ref[pad] == 0) and np.all(ref[1:][~pad[1:]] == 1)
    for f in ("x", "y", "z", "m", "s"):
        assert not np.any(rec[f][pad])
    # body ranges: leaves cover [0, n) exactly once
    leaves = np.flatnonzero((kind != 1) & ~pad)
    lo = first[leaves].astype(np.int64)
    cnt = np.where(kind[leaves] == 0, 1, count[leaves]).astype(np.int64)
    order = np.argsort(lo)
    lo, cnt = lo[order], cnt[order]
    assert lo[0] == 0 and np.array_equal(lo[1:], (lo + cnt)[:-1]) and lo[-1] + cnt[-1] == n
    # child edges are strictly smaller than the parent's
    for k in range(8):
        sel = count[internal] > k
        ch = starts[sel] + k
        cell = kind[ch] != 0
        assert np.all(s[ch][cell] < s[internal[sel]][cell])
    assert np.all(s[kind == 0] == -1.0)


@pytest.mark.parametrize("n,theta", CONFIGS)
def test_fullsize_properties(pkg, orc, n, theta):
    ic = pkg.plummer(n, seed=42)
    e = pkg.Engine(n, theta=theta, strict_fp=1)
    e.upload(*ic)
    e.tree_stages()
    keys = e.download_keys()
    assert np.all(keys[1:] >= keys[:-1])                                  # sortedness
    order = e.download_order()
    seen = np.zeros(n, np.bool_)
    seen[order] = True
    assert seen.all()                                                      # a permutation
    bodies = e.download_sorted_bodies()
    assert np.array_equal(bodies[:, 0], ic[0][order]) and np.array_equal(bodies[:, 3], ic[6][order])
    rec = e.download_tree()
    st = e.stats()
    assert st.status_flags == 0 and st.n_entries == len(rec)
    assert st.n_internal == int((rec["kind"] == 1).sum())
    _tree_invariants_vectorised(rec, n)
    m64 = ic[6].astype(np.float64)
    M = m64.sum()
    assert abs(float(rec["m"][0]) - M) <= 1e-6 * M
    for f, a in (("x", ic[0]), ("y", ic[1]), ("z", ic[2])):
        want = (m64 * a.astype(np.float64)).sum() / M
        assert abs(float(rec[f][0]) - want) <= 1e-5 * 400.0
    # strict kernel vs the oracle on the GPU's own tree (bit-exact, counters too): over ALL bodies up to 1M
    # (the oracle's 1M force pass costs about half a second on the box's cores), on a 4,096-body slab at 8M
    full = n <= 1_000_000
    lo, hi = (0, n) if full else ((n // 2) // 64 * 64, (n // 2) // 64 * 64 + 4096)
    e.force_count()
    gV, gO, gP = e.download_counters()
    ax, ay, az = e.download_acc()
    p = oparams(orc, e.params)
    oacc, V, O, P = orc.force(rec, bodies, p, orc.ORDER_BATCHED, lo, hi)
    sel = order[lo:hi]
    assert np.array_equal(ax[sel], oacc[lo:hi, 0])
    assert np.array_equal(ay[sel], oacc[lo:hi, 1])
    assert np.array_equal(az[sel], oacc[lo:hi, 2])
    assert np.array_equal(gV[sel], V[lo:hi]) and np.array_equal(gO[sel], O[lo:hi]) and np.array_equal(gP[sel], P[lo:hi])
    strict = np.stack([ax, ay, az], 1)
    e.close()
    # the FAST (benchmarked) kernel against the ORACLE ITSELF over the same bodies (all of them up to 1M).
    # Stated fp32 tolerance, as a distribution (a MAC decision can flip on a 1-ulp tie among ~1e9-1e10 decisions;
    # the flipped cell then differs by one cell's Barnes-Hut truncation error, itself far below the method's
    # ~1e-3 error): relative |da| median <= 1.2e-6 (theta 0.5) / 2.3e-6 (theta 0.3), 99.99th percentile <= 1.8e-5 /
    # 3.6e-5, and a max PER CONFIGURATION — each <= 2x the value measured on MI355X in rounds 3-4 (p50 / p99.99 /
    # max: 500k 5.5e-7 / 8.2e-6 / 2.6e-4; 1M 5.8e-7 / 8.8e-6 / 9.5e-5; 1M theta 0.3 1.1e-6 / 1.8e-5 / 7.1e-5; 8M vs the
    # strict kernel 6.4e-7 / 1.0e-5 / 2.0e-4)
    t50, t9999 = (1.2e-6, 1.8e-5) if theta >= 0.5 else (2.3e-6, 3.6e-5)
    tmax = {(500_000, 0.5): 5.2e-4, (1_000_000, 0.5): 1.9e-4, (1_000_000, 0.3): 1.5e-4, (8_000_000, 0.5): 4.0e-4}[(n, theta)]
    f = pkg.Engine(n, theta=theta)
    f.upload(*ic)
    f.tree_stages(); f.force()
    fast = np.stack(f.download_acc(), 1)
    oall = oacc[lo:hi, :3]
    rel_o = np.linalg.norm(fast[sel] - oall, axis=1) / np.linalg.norm(oall, axis=1)
    print(f"n={n} theta={theta}: fast vs oracle over {hi - lo} bodies: p50 {np.median(rel_o):.2e} "
          f"p99.99 {np.percentile(rel_o, 99.99):.2e} max {rel_o.max():.2e}")
    assert np.median(rel_o) <= t50 and np.percentile(rel_o, 99.99) <= t9999 and rel_o.max() <= tmax
    if not full:
        # 8M: over ALL bodies against the strict kernel (== oracle arithmetic, proven on the slab)
        rel = np.linalg.norm(fast - strict, axis=1) / np.linalg.norm(strict, axis=1)
        print(f"n={n}: fast vs strict all bodies: p50 {np.median(rel):.2e} p99.99 {np.percentile(rel, 99.99):.2e} max {rel.max():.2e}")
        assert np.median(rel) <= t50 and np.percentile(rel, 99.99) <= t9999 and rel.max() <= tmax
    # one whole step keeps the body set intact
    f.integrate()
    f.step(2)
    x, y, z, vx, vy, vz = f.download()
    assert np.isfinite(x).all() and np.isfinite(vx).all()
    assert np.array_equal(f.download_mass(), ic[6])                        # ids still line up
    assert f.stats().status_flags == 0
    f.close()


@pytest.mark.parametrize("n,theta,K", [(500_000, 0.5, 10), (1_000_000, 0.5, 10), (1_000_000, 0.3, 10), (8_000_000, 0.5, 3)])
def test_fullsize_k_steps_vs_oracle(pkg, orc, n, theta, K):
    """K whole steps of the default (fast) engine vs the oracle's step loop from identical Plummer inputs
    (ref:255-283 stage order on both sides) at every BASELINE size and both thetas (8M: K = 3).
    Stated fp32 tolerance on the state after K steps, caller order, as a distribution over the bodies
    (coordinates reach ~4000, so one ulp of a position is up to 4.9e-4; a MAC tie that flips changes one body's
    acceleration by one cell's truncation error):
        |dx|: median <= 3.1e-5 (one ulp at |x| < 512), 99.99th percentile <= 1.3e-4, max <= 2.5e-4
        |dv|: median <= 1e-6, 99.99th percentile <= 2.3e-5, max <= 4.6e-5
    each <= 2x the largest value measured on MI355X in round 3 (p50 / p99.99 / max, K = 10: 500k |dx| 0 / 3.1e-5 /
    6.1e-5, |dv| 0 / 7.6e-6 / 1.3e-5; 1M 0 / 6.1e-5 / 1.2e-4, 1.5e-8 / 7.6e-6 / 2.3e-5; 1M theta 0.3 0 / 6.1e-5 /
    1.2e-4, 4.8e-7 / 1.1e-5 / 2.3e-5).  8M bodies, K = 3 (the sphere's outskirts reach |x| ~ 8000 and the cube is
    twice as large, so ties and ulps are coarser): |dx| max <= 4.9e-4, |dv| p99.99 <= 6.1e-5, max <= 4.3e-4
    (measured 0 / 6.1e-5 / 2.4e-4 and 0 / 3.1e-5 / 2.1e-4)."""
    ic = pkg.plummer(n, seed=42)
    e = pkg.Engine(n, theta=theta)
    e.upload(*ic)
    e.step(K)
    g = np.stack(e.download(), 1).astype(np.float64)
    assert e.stats().status_flags == 0
    e.close()
    o = orc.Oracle(n, theta=theta)
    o.upload(*ic)
    o.step(K, order=orc.ORDER_BATCHED)
    w = np.stack(o.download(), 1).astype(np.float64)
    o.close()
    dx = np.abs(g[:, :3] - w[:, :3]).max(axis=1)
    dv = np.abs(g[:, 3:] - w[:, 3:]).max(axis=1)
    print(f"n={n} theta={theta} K={K}: |dx| p50 {np.median(dx):.3e} p99.99 {np.percentile(dx, 99.99):.3e} max {dx.max():.3e}; "
          f"|dv| p50 {np.median(dv):.3e} p99.99 {np.percentile(dv, 99.99):.3e} max {dv.max():.3e}")
    big = n > 1_000_000
    assert np.median(dx) <= 3.1e-5 and np.percentile(dx, 99.99) <= 1.3e-4 and dx.max() <= (4.9e-4 if big else 2.5e-4)
    assert np.median(dv) <= 1e-6 and np.percentile(dv, 99.99) <= (6.1e-5 if big else 2.3e-5) and \
        dv.max() <= (4.3e-4 if big else 4.6e-5)
