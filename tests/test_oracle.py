"""CPU tests of the oracle itself (no GPU).

The reference ships no golden vectors and cannot be built here, so the
oracle is formally "parity unpinned".  These tests pin it against things that do not depend on
the oracle's own code: analytic answers, an fp64 direct sum, numpy re-derivations of the integer
stages, committed fixtures (tests/golden/), and the algebraic properties of the recurrence.
"""
import json
import os

import numpy as np
import pytest

from helpers import oracle_pipeline, special_ics

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _np_expand10(v):
    """bit j of v -> bit 3j (definition of the reference's expandBits, nbody_v5_bench.cu:42-49)"""
    r = np.zeros_like(v, dtype=np.uint32)
    for j in range(10):
        r |= ((v >> j) & 1).astype(np.uint32) << np.uint32(3 * j)
    return r


def test_bbox_matches_numpy(pkg, orc):
    x, y, z = pkg.plummer(5000, seed=1)[:3]
    b = orc.bbox(x, y, z)
    size = np.float32(max(x.max() - x.min(), y.max() - y.min(), z.max() - z.min()))
    assert b[0] == x.min() and b[1] == y.min() and b[2] == z.min()
    assert b[3] == np.float32(x.min() + size) and b[4] == np.float32(y.min() + size)
    assert b[5] == np.float32(z.min() + size)


def test_morton30_matches_numpy_definition(pkg, orc):
    """reference-literal 30-bit code (nbody_v5_bench.cu:51-63) re-derived with numpy fp32 ops"""
    x, y, z = pkg.plummer(4096, seed=2)[:3]
    b = orc.bbox(x, y, z)
    codes, idx = orc.morton30(x, y, z, b)
    size = np.float32(max(b[3] - b[0], np.float32(1.0)))
    f = np.float32
    q = [((c - mn) / size * f(1023.0)).astype(np.uint32) for c, mn in ((x, b[0]), (y, b[1]), (z, b[2]))]
    want = (_np_expand10(q[0]) << np.uint32(2)) | (_np_expand10(q[1]) << np.uint32(1)) | _np_expand10(q[2])
    assert np.array_equal(codes, want)
    assert np.array_equal(idx, np.arange(4096))
    # the generic key path in 30-bit mode is the same code
    assert np.array_equal(orc.keys(x, y, z, b, 30).astype(np.uint32), codes)


def test_keys63_digits_are_octants(pkg, orc):
    """each octal digit of the 63-bit key is (xbit<<2 | ybit<<1 | zbit) of the 21-bit cell index"""
    x, y, z = pkg.plummer(3000, seed=3)[:3]
    b = orc.bbox(x, y, z)
    k = orc.keys(x, y, z, b, 63)
    size = np.float32(max(b[3] - b[0], np.float32(1.0)))
    f = np.float32
    q = [np.minimum(((c - mn) / size * f(2097152.0)).astype(np.uint32), 2097151)
         for c, mn in ((x, b[0]), (y, b[1]), (z, b[2]))]
    for level in range(21):
        sh = 20 - level
        want = (((q[0] >> sh) & 1) << 2) | (((q[1] >> sh) & 1) << 1) | ((q[2] >> sh) & 1)
        got = (k >> np.uint64(3 * sh)) & np.uint64(7)
        assert np.array_equal(got.astype(np.uint32), want.astype(np.uint32)), level


def test_sort_is_stable(orc):
    rng = np.random.default_rng(4)
    k = rng.integers(0, 50, 20000).astype(np.uint64)  # many ties
    sk, perm = orc.sort(k)
    want = np.argsort(k, kind="stable")
    assert np.array_equal(perm, want)
    assert np.array_equal(sk, k[want])


def _tree_invariants(rec, lo, hi, n, cap, compress):
    assert lo[0] == 0 and hi[0] == n
    seen_body = np.zeros(n, np.int32)
    internal = np.flatnonzero(rec["kind"] == 1)
    referenced = np.zeros(len(rec), np.int32)
    referenced[0] = 1
    for e in internal:
        f, c = rec["first"][e], rec["count"][e]
        assert 1 <= c <= 8
        if compress:
            assert c >= 2  # every emitted cell branches
        assert f % 2 == 0 and f >= 2                                 # blocks start on 64-byte boundaries
        if c % 2:
            assert rec["kind"][f + c] == 3                           # odd block: one padding entry follows
        ch = np.arange(f, f + c)
        referenced[ch] += 1
        assert lo[ch[0]] == lo[e] and hi[ch[-1]] == hi[e]          # children partition the parent
        assert np.array_equal(hi[ch[:-1]], lo[ch[1:]])
        assert np.all(rec["s"][ch][rec["kind"][ch] != 0] < rec["s"][e])  # edges shrink
        assert hi[e] - lo[e] > cap
    pad = rec["kind"] == 3
    assert pad[1] and np.all(referenced[pad] == 0)                   # padding is never referenced
    for f in ("x", "y", "z", "m", "s"):
        assert not np.any(rec[f][pad])
    assert np.all(referenced[~pad] == 1)                             # a tree: every entry has one parent
    leaves = np.flatnonzero((rec["kind"] != 1) & ~pad)
    for e in leaves:
        seen_body[lo[e]:hi[e]] += 1
    assert np.all(seen_body == 1)                                    # every body in exactly one leaf
    if compress:
        assert len(internal) <= max(n - 1, 0) and len(rec) - int(pad.sum()) <= 2 * n and len(rec) <= 3 * n + 2


@pytest.mark.parametrize("compress", [0, 1])
@pytest.mark.parametrize("cap", [1, 4])
def test_tree_invariants(pkg, orc, compress, cap):
    for n in (1, 2, 17, 3000):
        ic = pkg.plummer(n, seed=5)
        p = orc.params(leaf_cap=cap, compress=compress)
        o = oracle_pipeline(orc, ic, p)
        _tree_invariants(o["rec"], o["er_lo"], o["er_hi"], n, cap, compress)
        assert abs(float(o["rec"]["m"][0]) - float(ic[6].astype(np.float64).sum())) <= 1e-6 * float(ic[6].sum())


@pytest.mark.parametrize("name", ["coincident", "collinear", "outlier", "pairs", "tiny", "grid", "zero_mass"])
def test_tree_invariants_edge_cases(orc, name):
    ic = special_ics(name, 500, np.random.default_rng(6))
    for compress in (0, 1):
        p = orc.params(compress=compress)
        if compress == 0 and name == "pairs":
            continue  # chains of ~17 cells per pair overflow the 3n+8 pool by design
        o = oracle_pipeline(orc, ic, p)
        _tree_invariants(o["rec"], o["er_lo"], o["er_hi"], 500, 1, compress)


def test_uncompressed_pairs_overflow_is_reported(orc):
    """the literal chain tree needs more than 3n entries on close pairs: the oracle says so"""
    ic = special_ics("pairs", 500, np.random.default_rng(6))
    with pytest.raises(RuntimeError):
        oracle_pipeline(orc, ic, orc.params(compress=0))


def test_iterative_walk_equals_recursive_walk(pkg, orc):
    """the pre-order walk the CPU baseline times (explicit frames) == plain recursion, bit for bit"""
    for n, theta in ((5000, 0.5), (20000, 0.3)):
        ic = pkg.plummer(n, seed=13)
        p = orc.params(theta=theta)
        o = oracle_pipeline(orc, ic, p)
        a = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
        b = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER_RECURSIVE)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


def test_path_compression_preserves_forces(pkg, orc):
    """compress=1 (engine tree) vs compress=0 (literal chain cells): same accepted bodies for every
    particle; bit-identical accelerations in pre-order; fewer MAC evaluations."""
    for n, theta in ((3000, 0.5), (20000, 0.3)):
        ic = pkg.plummer(n, seed=7)
        res = []
        for compress in (0, 1):
            p = orc.params(theta=theta, compress=compress)
            o = oracle_pipeline(orc, ic, p)
            res.append(orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER) + (o["n_internal"],))
        (a0, V0, O0, P0, ni0), (a1, V1, O1, P1, ni1) = res
        assert np.array_equal(a0, a1)
        assert np.array_equal(P0, P1)
        assert np.array_equal(V0 - O0, V1 - O1)       # same accepted cells
        assert V1.sum() <= V0.sum() and ni1 <= ni0


def test_traversal_orders_agree(pkg, orc):
    ic = pkg.plummer(5000, seed=8)
    p = orc.params()
    o = oracle_pipeline(orc, ic, p)
    a0, V0, O0, P0 = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    a1, V1, O1, P1 = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_BATCHED)
    assert np.array_equal(V0, V1) and np.array_equal(O0, O1) and np.array_equal(P0, P1)
    scale = np.abs(a0).max()
    assert np.abs(a0 - a1).max() <= 2e-5 * scale   # summation order only


def test_two_body_analytic(orc):
    """a = G m d / (d^2 + eps^2)^(3/2) (nbody_v5_bench.cu:205-213)"""
    p = orc.params()
    x = np.array([0.0, 30.0], np.float32); y = np.zeros(2, np.float32); z = np.zeros(2, np.float32)
    m = np.array([3.0, 5.0], np.float32)
    ic = (x, y, z, y, y, y, m)
    o = oracle_pipeline(orc, ic, p)
    acc, V, O, P = orc.force(o["rec"], o["xyzm"], p)
    r2 = 30.0 ** 2 + 50.0
    want0 = 0.5 * 5.0 * 30.0 / r2 ** 1.5
    want1 = -0.5 * 3.0 * 30.0 / r2 ** 1.5
    inv = np.argsort(o["perm"])
    a = acc[inv]
    assert abs(a[0, 0] - want0) <= 1e-6 * abs(want0)
    assert abs(a[1, 0] - want1) <= 1e-6 * abs(want1)
    assert np.all(a[:, 1:3] == 0)
    assert np.all(P == 2)  # itself (zero contribution) + the other body


def test_theta0_equals_direct_sum(pkg, orc):
    n = 1500
    ic = pkg.plummer(n, seed=9)
    p = orc.params(theta=0.0)
    o = oracle_pipeline(orc, ic, p)
    acc, V, O, P = orc.force(o["rec"], o["xyzm"], p)
    assert np.all(P == n)
    d = orc.direct_f64(o["xyzm"], p.G, p.eps2)
    rel = np.linalg.norm(acc[:, :3] - d, axis=1) / np.linalg.norm(d, axis=1)
    assert rel.max() <= 2e-5


def test_bh_error_vs_direct_shrinks_with_theta(pkg, orc):
    n = 8192
    ic = pkg.plummer(n, seed=10)
    med = []
    for theta in (0.8, 0.5, 0.3):
        p = orc.params(theta=theta)
        o = oracle_pipeline(orc, ic, p)
        acc, *_ = orc.force(o["rec"], o["xyzm"], p, hi=1024)
        d = orc.direct_f64(o["xyzm"], p.G, p.eps2, 0, 1024)
        rel = np.linalg.norm(acc[:1024, :3] - d, axis=1) / np.linalg.norm(d, axis=1)
        med.append(np.median(rel))
    assert med[0] > med[1] > med[2]
    assert med[1] < 5e-3


def test_integrate_formulae(orc):
    """kick-drift with speed clamp (nbody_v5_bench.cu:232-248), checked with numpy fp32 ops"""
    rng = np.random.default_rng(11)
    n = 1000
    f = np.float32
    xyzm = rng.normal(0, 100, (n, 4)).astype(f)
    vel = rng.normal(0, 300, (n, 3)).astype(f)   # some speeds above the 500 clamp
    acc = np.zeros((n, 4), f); acc[:, :3] = rng.normal(0, 50, (n, 3))
    p = orc.params()
    nx, nv = orc.integrate(xyzm, vel, acc, p)
    dt, vmax = f(0.02), f(500.0)
    v = (vel + acc[:, :3] * dt).astype(f)
    s2 = ((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(f) + v[:, 2] * v[:, 2]).astype(f)
    over = s2 > vmax * vmax
    assert over.any() and (~over).any()
    scale = (vmax / np.sqrt(s2[over]).astype(f)).astype(f)
    v[over] = (v[over] * scale[:, None]).astype(f)
    assert np.array_equal(nv, v)
    assert np.array_equal(nx[:, :3], (xyzm[:, :3] + v * dt).astype(f))
    assert np.array_equal(nx[:, 3], xyzm[:, 3])


def test_whole_step_equals_stage_composition(pkg, orc):
    n = 2000
    ic = pkg.plummer(n, seed=12)
    p = orc.params()
    o = oracle_pipeline(orc, ic, p)
    acc, *_ = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_BATCHED)
    nx, nv = orc.integrate(o["xyzm"], o["vel"], acc, p)
    st = orc.Oracle(n, p)
    st.upload(*ic)
    st.step(1, order=orc.ORDER_BATCHED)
    x, y, z, vx, vy, vz = st.download()
    assert np.array_equal(x[o["perm"]], nx[:, 0]) and np.array_equal(vz[o["perm"]], nv[:, 2])
    c = st.counts()
    assert c["n_internal"] == o["n_internal"] and c["n_entries"] == len(o["rec"])


def test_survey_counts_65536(pkg, orc):
    """SURVEY §8(d) probe figures for 65,536 Plummer, theta 0.5 (literal chain tree): nodes
    0.484 N, depth 13, V 1458, P 177, O 239 per particle (+-3 %)."""
    n = 65536
    ic = pkg.plummer(n, seed=42)
    p = orc.params(compress=0)
    o = oracle_pipeline(orc, ic, p)
    acc, V, O, P = orc.force(o["rec"], o["xyzm"], p)
    assert abs(o["n_internal"] / n - 0.484) < 0.01
    assert o["max_level"] in (12, 13, 14)
    assert abs(V.mean() / 1458 - 1) < 0.03
    assert abs(O.mean() / 239 - 1) < 0.03
    assert abs(P.mean() / 177 - 1) < 0.03


def test_golden_fixtures(pkg, orc):
    """committed vectors (tests/golden/make_golden.py): guards the oracle AND the IC generator
    against silent drift between rounds / machines"""
    path = os.path.join(GOLD, "plummer4096_seed42.npz")
    g = np.load(path)
    ic = pkg.plummer(4096, seed=42)
    for i, name in enumerate(["x", "y", "z", "vx", "vy", "vz", "m"]):
        assert np.array_equal(ic[i], g[name]), name
    p = orc.params()
    o = oracle_pipeline(orc, ic, p)
    assert np.array_equal(o["bounds"], g["bounds"])
    assert np.array_equal(o["sorted_keys"], g["sorted_keys"])
    assert np.array_equal(o["perm"], g["perm"])
    codes, _ = orc.morton30(*ic[:3], o["bounds"])
    assert np.array_equal(codes, g["morton30"])
    for f in ("kind", "first", "count"):
        assert np.array_equal(o["rec"][f], g["rec_" + f]), f
    assert np.array_equal(o["rec"]["s"], g["rec_s"])
    for f in ("x", "y", "z", "m"):
        assert np.array_equal(o["rec"][f], g["rec_" + f]), f
    acc, V, O, P = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    assert np.array_equal(acc, g["acc_preorder"])
    assert np.array_equal(V, g["V"]) and np.array_equal(O, g["O"]) and np.array_equal(P, g["P"])
    nx, nv = orc.integrate(o["xyzm"], o["vel"], acc, p)
    assert np.array_equal(nx, g["xyzm_after"]) and np.array_equal(nv, g["vel_after"])
    meta = json.load(open(os.path.join(GOLD, "plummer4096_seed42.json")))
    assert meta["n_internal"] == o["n_internal"] and meta["n_entries"] == len(o["rec"])


def test_hilbert_numbering_is_a_hilbert_curve(orc):
    """the oracle's Hilbert numbering of the 2^21-per-axis cell grid (key_curve = 1) has the curve's defining
    properties: a bijection (index -> cell -> index), consecutive indices are face neighbours (exactly one
    coordinate changes, by one cell), and every 3L-bit index prefix is one aligned cube of the level-L octree —
    the property the tree build relies on"""
    rng = np.random.default_rng(5)
    top = 1 << 63
    starts = [0, top - 4100, 1 << 30, (1 << 60) - 2050, 0x1249249249249249 - 2000] + \
             [int(v) for v in rng.integers(0, top - 5000, 12)]
    for s0 in starts:
        prev = orc.hilbert_cell(s0)
        assert orc.hilbert_index(*prev) == s0
        for i in range(s0 + 1, s0 + 4097):
            c = orc.hilbert_cell(i)
            assert orc.hilbert_index(*c) == i
            d = [abs(c[a] - prev[a]) for a in range(3)]
            assert sorted(d) == [0, 0, 1], (i, prev, c)
            prev = c
    for L in (1, 2, 5, 13, 20):
        sh = 3 * (21 - L)
        for pfx in [int(v) for v in rng.integers(0, 1 << (3 * L), 6)]:
            lo, hi = pfx << sh, ((pfx + 1) << sh) - 1
            probes = [lo, hi] + [int(v) for v in rng.integers(lo, hi + 1, 20)]
            cells = np.array([orc.hilbert_cell(q) for q in probes], dtype=np.int64)
            assert np.all((cells >> (21 - L)) == (cells[0] >> (21 - L)))   # one cube of edge 2^(21-L)


def test_hilbert_keys_sort_bodies_into_the_same_cells(pkg, orc):
    """key_curve changes the ORDER of the cells, not the cells: per octree level the multiset of cell populations
    is the same under Morton and Hilbert keys, and so is the canonical tree's size"""
    x, y, z, *_ = pkg.plummer(20000, seed=3)
    b = orc.bbox(x, y, z)
    km = orc.keys(x, y, z, b, 63, 0)
    kh = orc.keys(x, y, z, b, 63, 1)
    for L in (1, 3, 6, 10, 15, 21):
        sh = np.uint64(3 * (21 - L))
        cm = np.sort(np.unique(km >> sh, return_counts=True)[1])
        ch = np.sort(np.unique(kh >> sh, return_counts=True)[1])
        assert np.array_equal(cm, ch), L
    trees = []
    for kc, k in ((0, km), (1, kh)):
        sk, perm = orc.sort(k)
        p = orc.params(key_curve=kc)
        rec, lo, hi, ni, ml = orc.build_tree(sk, p, orc.root_edge(b))
        trees.append((len(rec), ni, ml))
    assert trees[0] == trees[1]
