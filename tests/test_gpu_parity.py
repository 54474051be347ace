"""GPU parity tests: every stage of the HIP path (through the C-ABI) against the CPU oracle
on identical seeded inputs.  Integer / index / byte results are compared bit for bit; floating
point tolerances are written next to each assert.

The reference (nbody_v5_bench.cu) has no tests of its own; the cases follow SURVEY.md §4.
"""
import numpy as np
import pytest

from helpers import key_curve_of, oracle_pipeline, oparams, special_ics

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 64, 65, 1000, 4096, 65536]


def _engine(pkg, ic, **kw):
    n = len(ic[0])
    e = pkg.Engine(n, **kw)
    e.upload(*ic)
    return e


@pytest.mark.parametrize("n", SIZES)
def test_bbox_bit_exact(pkg, orc, n):
    ic = pkg.plummer(n, seed=7)
    e = _engine(pkg, ic)
    e.bbox()
    gb = e.download_bounds()
    ob = orc.bbox(*ic[:3])
    assert gb.tobytes() == ob.tobytes()
    e.close()


@pytest.mark.parametrize("key_curve", [0, 1])
@pytest.mark.parametrize("key_bits", [63, 30])
@pytest.mark.parametrize("n", [1, 65, 4096, 65536])
def test_keys_bit_exact(pkg, orc, n, key_bits, key_curve):
    ic = pkg.plummer(n, seed=11)
    e = _engine(pkg, ic, key_bits=key_bits, max_depth=key_bits // 3, key_curve=key_curve)
    e.bbox(); e.morton()
    gk = e.download_keys()
    b = orc.bbox(*ic[:3])
    ok = orc.keys(*ic[:3], b, key_bits, key_curve_of(e.params))  # Morton and Hilbert (30-bit: always Morton)
    assert np.array_equal(gk, ok)
    if key_bits == 30:  # reference-literal Morton code (nbody_v5_bench.cu:42-63)
        codes, idx = orc.morton30(*ic[:3], b)
        assert np.array_equal(gk.astype(np.uint32), codes)
        assert np.array_equal(idx, np.arange(n))
    e.close()


@pytest.mark.parametrize("sort_variant", [0, 1, 2, 3])
@pytest.mark.parametrize("key_bits", [63, 30])
@pytest.mark.parametrize("n", [1, 2, 65, 4096, 4097, 65536, 300001])
def test_sort_stable_permutation(pkg, orc, n, key_bits, sort_variant):
    """every sort implementation (0 = automatic, 1 = histogram/scan/scatter radix passes, 2 = one kernel per
    radix pass with look-back, 3 = splitter sort forced on caller-order input) == the oracle's stable merge
    sort, exact permutation"""
    ic = pkg.plummer(n, seed=3)
    e = _engine(pkg, ic, key_bits=key_bits, max_depth=key_bits // 3, sort_variant=sort_variant)
    e.bbox(); e.morton(); e.sort()
    gk = e.download_keys()
    order = e.download_order()
    b = orc.bbox(*ic[:3])
    sk, perm = orc.sort(orc.keys(*ic[:3], b, key_bits, key_curve_of(e.params)))
    assert np.array_equal(gk, sk)                    # ascending keys
    assert np.array_equal(order, perm)               # stable: identical permutation
    bodies = e.download_sorted_bodies()
    assert np.array_equal(bodies[:, 0], ic[0][perm])  # physical gather
    assert np.array_equal(bodies[:, 3], ic[6][perm])
    assert e.stats().status_flags == 0
    e.close()


@pytest.mark.parametrize("sort_variant", [0, 1, 2, 3])
def test_sort_many_ties_and_repeated_calls(pkg, orc, sort_variant):
    """heavy ties (grid input, 30-bit keys) and 20 consecutive sorts on one context: the look-back
    table is never cleared between calls (tagged granules, monotonic tickets)"""
    ic = special_ics("grid", 50000, np.random.default_rng(3))
    e = _engine(pkg, ic, key_bits=30, max_depth=10, sort_variant=sort_variant)
    b = orc.bbox(*ic[:3])
    sk, perm = orc.sort(orc.keys(*ic[:3], b, 30))
    for it in range(20):
        e.upload(*ic)
        e.bbox(); e.morton(); e.sort()
        assert np.array_equal(e.download_order(), perm), it
        assert np.array_equal(e.download_keys(), sk), it
    assert e.stats().status_flags == 0
    e.close()


@pytest.mark.parametrize("n,shuffle", [(1000000, True), (1000000, False), (1500000, False), (200000, True)])
def test_splitter_sort_bucket_paths(pkg, orc, n, shuffle):
    """splitter sort forced (sort_variant 3): on caller-order (random) input a good part of the 256 buckets
    exceeds the 8192 keys that fit LDS and goes through the one-workgroup global-memory path; on input that is
    already in key order every bucket takes the LDS path.  Both == the oracle's stable sort."""
    ic = pkg.plummer(n, seed=11)
    b = orc.bbox(*ic[:3])
    kc = key_curve_of(pkg.default_params())
    keys = orc.keys(*ic[:3], b, 63, kc)
    if not shuffle:  # present the bodies in key order, as a step leaves them
        _, p0 = orc.sort(keys)
        ic = tuple(a[p0] for a in ic)
        keys = orc.keys(*ic[:3], b, 63, kc)
    e = _engine(pkg, ic, sort_variant=3)
    e.bbox(); e.morton(); e.sort()
    sk, perm = orc.sort(keys)
    assert np.array_equal(e.download_keys(), sk)
    assert np.array_equal(e.download_order(), perm)
    bodies = e.download_sorted_bodies()
    assert np.array_equal(bodies[:, 0], ic[0][perm])
    assert e.stats().status_flags == 0
    e.close()


@pytest.mark.parametrize("n,presorted", [(30000, False), (30000, True), (400000, True)])
def test_splitter_sort_close_clusters(pkg, orc, n, presorted):
    """the in-LDS bucket sort (ls_sort_in_lds) has four paths: the full set of radix passes at once (the input's
    adjacent keys say a window would leave many ties), passes over a window of the top 24-48 bits only, the same
    followed by neighbour exchanges inside the runs of keys that agree on the window, and — exchanges not finished —
    the full set after all.  Input made for all of them (tools/lsort_paths.py shows which bucket took which):
    scattered bodies, 300 pairs and 40 clumps of 5-30 bodies 1e-5 apart, clumps of 700 bodies 1e-4 ... 0.5 wide,
    exact duplicates (equal keys keep their original order), 64 clumps dealt round-robin into the input (no tie
    between adjacent INPUT bodies, runs of 150 in the result).  == the oracle's stable sort, exact permutation."""
    rng = np.random.default_rng(17)
    x = rng.uniform(-1000.0, 1000.0, (n, 3))
    k = 1000
    for _ in range(300):  # pairs
        x[k + 1] = x[k] + rng.normal(0, 1e-5, 3)
        k += 2
    for _ in range(40):   # small clumps
        m = int(rng.integers(5, 31))
        x[k:k + m] = x[k] + rng.normal(0, 1e-5, (m, 3))
        k += m
    for wdt in (0.5, 0.1, 0.02, 1e-4):  # big clumps, wider than a finest cell: hundreds of distinct keys per run
        x[k:k + 700] = x[k] + rng.uniform(-wdt, wdt, (700, 3))
        k += 700
    x[k:k + 50] = x[k]  # coincident bodies
    k += 50
    # 64 clumps of 150 bodies, each 2e-3 wide, dealt ROUND-ROBIN into the input: adjacent input bodies always
    # belong to different clumps (the window estimate sees no ties), the sorted buckets hold runs of 150 keys in
    # random order, more than the neighbour exchanges finish: the window passes are followed by the full set
    cc = rng.uniform(-900.0, 900.0, (64, 3))
    rr = cc[np.arange(9600) % 64] + rng.uniform(-1e-3, 1e-3, (9600, 3))
    x[k:k + 9600] = rr
    x = x.astype(np.float32)
    z = np.zeros(n, np.float32)
    ic = (x[:, 0].copy(), x[:, 1].copy(), x[:, 2].copy(), z, z.copy(), z.copy(), np.ones(n, np.float32))
    b = orc.bbox(*ic[:3])
    kc = key_curve_of(pkg.default_params())
    keys = orc.keys(*ic[:3], b, 63, kc)
    if presorted:  # bodies in key order except for a sprinkling of swaps, as a step leaves them
        _, p0 = orc.sort(keys)
        sw = rng.integers(0, n - 40, 200)
        p0[sw], p0[sw + 37] = p0[sw + 37].copy(), p0[sw].copy()
        ic = tuple(a[p0] for a in ic)
        keys = orc.keys(*ic[:3], b, 63, kc)
    e = _engine(pkg, ic, sort_variant=3)
    e.bbox(); e.morton(); e.sort()
    sk, perm = orc.sort(keys)
    assert np.array_equal(e.download_keys(), sk)
    assert np.array_equal(e.download_order(), perm)
    st = e.stats()
    assert st.status_flags == 0
    if presorted:
        assert st.sort_slow_buckets == 0  # every bucket took the in-LDS path under test
    e.close()


def test_splitter_sort_gives_way_when_buckets_overflow(pkg, orc):
    """many identical keys (120,000 coincident bodies among 200,000) land in ONE bucket of the splitter sort, far
    beyond the 8,192 keys its workgroup sorts in LDS: that bucket goes through the slow global-memory path (still
    the exact stable order), bh_stats.sort_slow_buckets reports it, and once bh_get_stats has seen it the context
    sorts with the radix passes until the next upload — results identical either way"""
    n = 200000
    x, y, z, vx, vy, vz, m = [a.copy() for a in pkg.plummer(n, seed=23)]
    x[:120000] = 1.5; y[:120000] = -2.5; z[:120000] = 3.25
    vx[:120000] = vy[:120000] = vz[:120000] = 0.0
    ic = (x, y, z, vx, vy, vz, m)
    e = _engine(pkg, ic)
    e.step(2)                       # radix (first sort), then the splitter sort meets the overfull bucket
    st = e.stats()
    assert st.status_flags == 0 and st.sort_slow_buckets >= 1
    slow = st.sort_slow_buckets
    e.step(3)                       # bh_get_stats saw the counter: radix passes from now on
    st = e.stats()
    assert st.status_flags == 0 and st.sort_slow_buckets == slow
    a = _state(e)
    e.close()
    r = _engine(pkg, ic, sort_variant=2)   # pinned to the radix sort
    r.step(5)
    assert _state(r) == a
    r.close()


def test_splitter_sort_gives_way_in_a_plain_step_loop(pkg):
    """the same input in the loop INTEGRATION.md shows — bh_step + bh_sync per frame, bh_get_stats never polled in
    between (round-3 review: the fall-back was only evaluated inside bh_get_stats): bh_sync reads the slow-bucket
    count with the sticky flags, so the overfull bucket is met once, not in every step"""
    n = 200000
    x, y, z, vx, vy, vz, m = [a.copy() for a in pkg.plummer(n, seed=23)]
    x[:120000] = 1.5; y[:120000] = -2.5; z[:120000] = 3.25
    vx[:120000] = vy[:120000] = vz[:120000] = 0.0
    e = _engine(pkg, (x, y, z, vx, vy, vz, m))
    for _ in range(6):
        e.step(); e.sync()
    st = e.stats()
    assert st.status_flags == 0 and 1 <= st.sort_slow_buckets <= 2   # step 2 (and at most the step in flight)
    e.close()


def test_splitter_sort_in_steps(pkg, orc):
    """automatic choice: the first sort after an upload is the radix sort, later steps use the splitter sort;
    the body order after 6 steps equals that of a context pinned to the radix sort, bit for bit"""
    ic = pkg.plummer(120000, seed=5)
    out = []
    for sv in (0, 2):
        e = _engine(pkg, ic, sort_variant=sv)
        for _ in range(6):
            e.step()
        out.append((e.download_order(), e.download_sorted_bodies(), e.download_keys()))
        assert e.stats().status_flags == 0
        e.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1].tobytes() == out[1][1].tobytes()
    assert np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("n", [256 * 6144, 256 * 6144 + 1])
def test_splitter_sort_size_limit(pkg, n):
    """256 buckets x 6144 bodies is the largest context the automatic choice gives to the splitter sort; one body
    more and every step uses the radix sort.  Either way == a context pinned to the radix sort, bit for bit."""
    ic = pkg.plummer(n, seed=8)
    out = []
    for sv in (0, 2):
        e = _engine(pkg, ic, sort_variant=sv)
        e.step(3)
        out.append((e.download_order().tobytes(), e.download_keys().tobytes()))
        assert e.stats().status_flags == 0
        e.close()
    assert out[0] == out[1]


@pytest.mark.parametrize("n", [5000, 200000])
def test_force_placement_and_group_size_do_not_change_results(pkg, n):
    """bh_params.xcd_mode (workgroup -> body-chunk placement) and force_group (bodies per wave) are speed knobs of
    the one-wave-per-group walk (force_coop = 1): accelerations are bit-identical for every setting.  (With several
    waves per group the placement still changes nothing, the group composition does: test_force_coop_*.)"""
    ic = pkg.plummer(n, seed=6)
    ref = None
    for kw in (dict(), dict(xcd_mode=0), dict(xcd_mode=1), dict(xcd_mode=2), dict(force_group=16),
               dict(force_group=32), dict(force_group=64), dict(force_block=256, xcd_mode=2)):
        e = _engine(pkg, ic, force_coop=1, **kw)
        e.tree_stages(); e.force()
        a = np.stack(e.download_acc(), 1).tobytes()
        assert e.stats().status_flags == 0
        e.close()
        if ref is None:
            ref = a
        assert a == ref, kw


@pytest.mark.parametrize("n", [1, 2, 63, 65, 1000, 4096, 24577, 65536, 200000])
def test_force_coop_walk_matches_the_one_wave_walk(pkg, n):
    """K waves share the walk of one group level by level (force_coop_kernel, bh_params.force_coop = K): every body
    meets exactly the records it meets in the one-wave walk — the accept / open decisions are per body — but its
    accepted records are summed in K partial sums added in wave order.  Against the one-wave walk (force_coop = 1):
    relative |da| per body median <= 3e-7, 99.99th percentile <= 1e-5, max <= 2e-4 (fp32 association only; the
    maximum belongs to bodies near the centre whose large partial sums cancel: 3.7e-5 measured at 200,000 bodies in
    round 4, profiles/r04_parity/coop.txt).  Reproducible: the same K and group size give the same bits on a second context,
    and the placement (xcd_mode) changes nothing."""
    ic = pkg.plummer(n, seed=6)
    e = _engine(pkg, ic, force_coop=1)
    e.tree_stages(); e.force()
    ref = np.stack(e.download_acc(), 1).astype(np.float64)
    e.close()
    norm = np.maximum(np.sqrt((ref ** 2).sum(1)), 1e-30)
    worst = 0.0
    for K, group in ((2, 64), (3, 64), (4, 64), (8, 64), (4, 32), (8, 32), (5, 16)):
        outs = []
        for kw in (dict(), dict(xcd_mode=1), dict(xcd_mode=2)):
            e = _engine(pkg, ic, force_coop=K, force_group=group, **kw)
            e.tree_stages(); e.force()
            outs.append(np.stack(e.download_acc(), 1))
            st = e.stats()
            assert st.status_flags == 0 and st.force_redo_waves == 0, (K, group, kw)
            e.close()
        assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes(), (K, group)
        rel = np.sqrt(((outs[0].astype(np.float64) - ref) ** 2).sum(1)) / norm
        worst = max(worst, float(rel.max()))
        assert np.median(rel) <= 3e-7 and np.percentile(rel, 99.99) <= 1e-5 and rel.max() <= 2e-4, \
            (K, group, float(np.median(rel)), float(np.percentile(rel, 99.99)), float(rel.max()))
    print(f"coop vs one-wave walk n={n}: worst relative |da| {worst:.3e}")


@pytest.mark.parametrize("name", ["coincident", "collinear", "outlier", "pairs", "tiny", "grid", "zero_mass"])
def test_force_coop_walk_edge_inputs(pkg, name):
    """the cooperative walk on the edge inputs: coincident bodies form unsplit cells of more than 8 bodies
    (the group is then redone by wave 0 with the generic loop, same results), chains of close
    pairs are deep, zero masses are skipped.  Against the one-wave walk, same bound as above; leaf_cap 4 as well."""
    rng = np.random.default_rng(5)
    ic = special_ics(name, 3000, rng)
    for leaf_cap in (1, 4):
        e = _engine(pkg, ic, force_coop=1, leaf_cap=leaf_cap)
        e.tree_stages(); e.force()
        ref = np.stack(e.download_acc(), 1).astype(np.float64)
        assert e.stats().status_flags == 0
        e.close()
        scale = max(float(np.sqrt((ref ** 2).sum(1)).max()), 1e-30)
        for K in (2, 4, 7):
            e = _engine(pkg, ic, force_coop=K, force_group=64, leaf_cap=leaf_cap)
            e.tree_stages(); e.force()
            a = np.stack(e.download_acc(), 1).astype(np.float64)
            assert e.stats().status_flags == 0
            e.close()
            assert np.isfinite(a).all()
            assert np.abs(a - ref).max() <= 2e-5 * scale, (name, leaf_cap, K)


def test_force_coop_unsplit_cells_of_thousands_of_bodies(pkg):
    """a depth cap of 5 leaves unsplit cells of several thousand bodies; their child count (the walk keeps the largest
    it met to detect blocks of more than 8 children) must not read as one of the walk's flag bits — found by
    tools/coop_fuzz.py in round 4: a count above 2,047 looked like "traversal limit" and the group kept partial sums.
    Such groups are redone by wave 0 with the generic loop: same accelerations as the one-wave walk, no flag."""
    n = 88123
    ic = pkg.plummer(n, seed=5)
    e = _engine(pkg, ic, force_coop=1, max_depth=5, theta=0.8)
    e.tree_stages(); e.force()
    ref = np.stack(e.download_acc(), 1).astype(np.float64)
    assert e.stats().status_flags == 0
    e.close()
    for K in (2, 4):
        e = _engine(pkg, ic, force_coop=K, max_depth=5, theta=0.8)
        e.tree_stages(); e.force()
        a = np.stack(e.download_acc(), 1).astype(np.float64)
        st = e.stats()
        e.close()
        assert st.status_flags == 0 and st.force_redo_waves > 0
        rel = np.sqrt(((a - ref) ** 2).sum(1)) / np.sqrt((ref ** 2).sum(1))
        assert rel.max() <= 2e-4, (K, float(rel.max()))


@pytest.mark.parametrize("theta", [0.3, 0.1])
def test_force_coop_full_level_lists_spill_to_the_wave_stack(pkg, theta):
    """theta = 0.3 / 0.1 open far more cells per level than a wave's level list holds (127 entries; the longest list
    at theta 0.3 is ~170, tools/coop_lists.py): what a full list cannot take goes onto the wave's own cross-lane stack
    and is walked depth-first by that wave (BH_PUSH1, coop branch) — nothing is redone, nothing is lost: against the
    one-wave walk relative |da| median <= 3e-6, max <= 2e-4 (summation order of thousands of terms per body)."""
    n = 20000
    ic = pkg.plummer(n, seed=8)
    e = _engine(pkg, ic, force_coop=1, theta=theta)
    e.tree_stages(); e.force()
    ref = np.stack(e.download_acc(), 1).astype(np.float64)
    e.close()
    for K in (2, 4):
        e = _engine(pkg, ic, force_coop=K, force_group=64, theta=theta)
        e.tree_stages(); e.force()
        a = np.stack(e.download_acc(), 1).astype(np.float64)
        st = e.stats()
        e.close()
        assert st.status_flags == 0 and st.force_redo_waves == 0
        rel = np.sqrt(((a - ref) ** 2).sum(1)) / np.sqrt((ref ** 2).sum(1))
        assert np.median(rel) <= 3e-6 and rel.max() <= 2e-4, (K, float(np.median(rel)), float(rel.max()))


@pytest.mark.parametrize("n", [5000, 70000, 400000])
def test_force_launch_trace_is_the_product_launch(pkg, n):
    """bh_force_launch_trace (measurement: bench.py's roofline.issue.residency) runs the traced instance of the launch
    bh_step makes — cooperative throughout (5,000 / 70,000 bodies: 8 / 4 waves per group) or mixed (400,000) — and
    stores the same accelerations as bh_force, bit for bit; one row per wave, every SIMD of the GPU seen at the larger
    sizes, no wave ends before it starts"""
    ic = pkg.plummer(n, seed=4)
    e = _engine(pkg, ic)
    e.tree_stages(); e.force()
    a = np.stack(e.download_acc(), 1)
    rows = e.force_launch_trace()
    b = np.stack(e.download_acc(), 1)
    assert a.tobytes() == b.tobytes()
    groups = (n + 63) // 64
    tail = 256 * 4 * 7 // 3
    if groups * 8 <= 256 * 4 * 7:
        want = groups * 8
    elif groups <= 2 * tail:
        want = groups * 4
    else:
        gb = (groups - tail) // 4 * 4
        want = gb + 4 * (groups - gb)
    assert rows.shape == (want, 4), (rows.shape, want)
    dt = (rows[:, 1].astype(np.int64) - rows[:, 0].astype(np.int64))
    assert (dt >= 0).all() and dt.max() < 100_000_000
    assert e.stats().status_flags == 0
    e.close()
    s = _engine(pkg, ic, strict_fp=1)
    s.tree_stages()
    assert len(s.force_launch_trace()) == 0      # no traced instance of the strict walk
    s.close()


def _check_tree(pkg, orc, ic, **kw):
    e = _engine(pkg, ic, **kw)
    e.tree_stages()
    rec = e.download_tree()
    st = e.stats()
    p = oparams(orc, e.params)
    o = oracle_pipeline(orc, ic, p)
    orec = o["rec"]
    assert st.status_flags == 0
    assert len(rec) == len(orec) == st.n_entries
    assert st.n_internal == o["n_internal"]
    assert st.max_level == o["max_level"]
    for f in ("kind", "first", "count"):
        assert np.array_equal(rec[f], orec[f]), f
    assert rec["s"].tobytes() == orec["s"].tobytes()
    # centres of mass: GPU = fp64 prefix differences, oracle = fp64 per-cell sums, both rounded
    # once to fp32 -> agree to 2 ulp of the coordinate scale; masses to 1 ulp
    scale = max(float(np.abs(o["xyzm"][:, :3]).max()), 1.0)
    tol = 4 * np.finfo(np.float32).eps * scale
    for f in ("x", "y", "z"):
        assert np.abs(rec[f] - orec[f]).max() <= tol, f
    assert np.all(np.abs(rec["m"] - orec["m"]) <= 2 * np.finfo(np.float32).eps * np.abs(orec["m"]))
    body = rec["kind"] == pkg.KIND_BODY
    for f in ("x", "y", "z", "m"):  # body records are copies: exact
        assert np.array_equal(rec[f][body], orec[f][body])
    return e, rec, o, p


@pytest.mark.parametrize("key_curve", [0, 1])
@pytest.mark.parametrize("n", SIZES)
def test_tree_topology_and_com(pkg, orc, n, key_curve):
    e, *_ = _check_tree(pkg, orc, pkg.plummer(n, seed=5), key_curve=key_curve)
    e.close()


@pytest.mark.parametrize("kw", [dict(leaf_cap=4), dict(leaf_cap=16), dict(max_depth=3),
                                dict(max_depth=0), dict(key_bits=30, max_depth=10),
                                dict(key_bits=30, max_depth=10, leaf_cap=8)])
def test_tree_variants(pkg, orc, kw):
    e, *_ = _check_tree(pkg, orc, pkg.plummer(5000, seed=9), **kw)
    e.close()


@pytest.mark.parametrize("name", ["coincident", "collinear", "outlier", "pairs", "tiny", "grid", "zero_mass"])
def test_tree_edge_cases(pkg, orc, name):
    ic = special_ics(name, 777, np.random.default_rng(1))
    e, *_ = _check_tree(pkg, orc, ic)
    e.close()


def _strict_force_check(pkg, orc, ic, **kw):
    """strict_fp kernel vs the oracle walking THE GPU'S OWN tree in the same (batched) order:
    identical interactions in identical order with identical IEEE arithmetic -> bit-exact."""
    e = _engine(pkg, ic, strict_fp=1, **kw)
    e.tree_stages()
    e.force_count()
    rec = e.download_tree()
    bodies = e.download_sorted_bodies()
    order = e.download_order()
    p = oparams(orc, e.params)
    oacc, V, O, P = orc.force(rec, bodies, p, orc.ORDER_BATCHED)
    ax, ay, az = e.download_acc()
    gV, gO, gP = e.download_counters()
    assert np.array_equal(ax[order], oacc[:, 0])
    assert np.array_equal(ay[order], oacc[:, 1])
    assert np.array_equal(az[order], oacc[:, 2])
    assert np.array_equal(gV[order], V)
    assert np.array_equal(gO[order], O)
    assert np.array_equal(gP[order], P)
    st = e.stats()
    assert st.count_V == int(V.sum()) and st.count_O == int(O.sum()) and st.count_P == int(P.sum())
    assert st.status_flags == 0
    return e, oacc, order


@pytest.mark.parametrize("key_curve", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 65, 1000, 4096, 65536])
def test_force_strict_bit_exact(pkg, orc, n, key_curve):
    e, *_ = _strict_force_check(pkg, orc, pkg.plummer(n, seed=42), key_curve=key_curve)
    e.close()


@pytest.mark.parametrize("theta", [0.0, 0.3, 1.0])
def test_force_strict_theta(pkg, orc, theta):
    e, *_ = _strict_force_check(pkg, orc, pkg.plummer(3000, seed=2), theta=theta)
    e.close()


@pytest.mark.parametrize("kw", [dict(leaf_cap=8), dict(max_depth=4), dict(key_bits=30, max_depth=10)])
def test_force_strict_variants(pkg, orc, kw):
    e, *_ = _strict_force_check(pkg, orc, pkg.plummer(5000, seed=4), **kw)
    e.close()


@pytest.mark.parametrize("name", ["coincident", "collinear", "outlier", "pairs", "tiny", "grid", "zero_mass"])
def test_force_strict_edge_cases(pkg, orc, name):
    e, *_ = _strict_force_check(pkg, orc, special_ics(name, 777, np.random.default_rng(1)))
    e.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", ["coincident", "collinear", "outlier", "pairs", "tiny", "grid", "zero_mass"])
def test_force_fast_edge_cases(pkg, orc, name, variant):
    """both fast kernels (0 = hand-scheduled walk, 1 = compiler-scheduled walk) on the edge inputs,
    incl. deep multi-body leaves (coincident) and mass<=0 records; tolerance as below"""
    ic = special_ics(name, 777, np.random.default_rng(1))
    e = _engine(pkg, ic, force_variant=variant)
    e.tree_stages(); e.force()
    ga = np.stack(e.download_acc(), 1)
    p = oparams(orc, e.params)
    o = oracle_pipeline(orc, ic, p)
    oacc, *_ = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    oa = np.zeros((777, 3), np.float32)
    oa[o["perm"]] = oacc[:, :3]
    scale = np.linalg.norm(oa, axis=1).max()
    assert np.isfinite(ga).all()
    assert np.linalg.norm(ga - oa, axis=1).max() <= 2e-4 * max(scale, 1e-30)
    assert e.stats().status_flags == 0
    e.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n", [777, 5001, 5002])
@pytest.mark.parametrize("kw", [dict(leaf_cap=4), dict(leaf_cap=8), dict(max_depth=3), dict(leaf_cap=3, max_depth=5)])
def test_force_fast_unsplit_cells_odd_and_even_n(pkg, orc, kw, n, variant):
    """unsplit multi-body cells (leaf_cap > 1 / depth cap): their bodies' digests form child blocks in the second
    region of the digest pool, which must start at EVEN records whatever the parity of n (round 2 had
    rec_cap = 3n + 8: odd for odd n, every such block then straddled two 64-byte pairs and its bodies exerted
    no force).  Both fast walks vs the oracle; a body with all its neighbours missing would be off by O(1)."""
    ic = pkg.plummer(n, seed=9)
    e = _engine(pkg, ic, force_variant=variant, **kw)
    e.tree_stages(); e.force()
    ga = np.stack(e.download_acc(), 1)
    rec = e.download_tree()
    assert (rec["kind"] == orc.KIND_MULTI).sum() > 0   # the case under test is present
    p = oparams(orc, e.params)
    o = oracle_pipeline(orc, ic, p)
    oacc, *_ = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    oa = np.zeros((n, 3), np.float32)
    oa[o["perm"]] = oacc[:, :3]
    rel = np.linalg.norm(ga - oa, axis=1) / np.linalg.norm(oa, axis=1)
    assert np.median(rel) <= 2e-6 and rel.max() <= 5e-4, (np.median(rel), rel.max())
    st = e.stats()
    assert st.status_flags == 0
    e.close()


@pytest.mark.parametrize("theta", [0.0, 0.2, 0.5, 1.0])
def test_force_block_sizes_agree(pkg, orc, theta):
    """the wave is the unit of work: 64-, 128- and 256-thread workgroups give identical results"""
    n = 30000
    ic = pkg.plummer(n, seed=21)
    acc = []
    for fb in (64, 128, 256):
        e = _engine(pkg, ic, theta=theta, force_block=fb)
        e.tree_stages(); e.force()
        acc.append(np.stack(e.download_acc(), 1))
        assert e.stats().status_flags == 0
        e.close()
    assert np.array_equal(acc[0], acc[1]) and np.array_equal(acc[0], acc[2])


def test_force_variant_out_of_range(pkg):
    with pytest.raises(Exception):
        pkg.Engine(1000, force_variant=2)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n,theta", [(4096, 0.5), (65536, 0.5), (65536, 0.3)])
def test_force_fast_vs_oracle(pkg, orc, n, theta, variant):
    """Default (fast) kernel: fma + v_rsq_f32 instead of sqrtf and '/'.  Stated fp32 tolerance:
    relative deviation |a_gpu - a_oracle| / |a_oracle| has median <= 2e-6, 99.9th percentile <= 2e-5 and
    max <= 5e-4 (a MAC decision can flip on a 1-ulp tie; the flipped cell then differs by the
    Barnes-Hut truncation error of one cell — 2.2e-4 is the largest seen over the sizes, thetas and both key
    curves here — still below the method's own ~1e-3 error)."""
    ic = pkg.plummer(n, seed=42)
    e = _engine(pkg, ic, theta=theta, force_variant=variant)
    e.tree_stages(); e.force()
    ax, ay, az = e.download_acc()
    p = oparams(orc, e.params)
    o = oracle_pipeline(orc, ic, p)
    oacc, *_ = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    oa = np.zeros((n, 3), np.float32)
    oa[o["perm"]] = oacc[:, :3]
    ga = np.stack([ax, ay, az], 1)
    rel = np.linalg.norm(ga - oa, axis=1) / np.linalg.norm(oa, axis=1)
    assert np.median(rel) <= 2e-6
    assert np.quantile(rel, 0.999) <= 2e-5
    assert rel.max() <= 5e-4
    e.close()


@pytest.mark.parametrize("n", [125_000, 250_000])
def test_force_default_engine_vs_oracle_where_every_group_is_cooperative(pkg, orc, n):
    """125,000 (the strong-scaling share of 1M over 8 GPUs) and 250,000 bodies: the default engine walks EVERY group
    with four waves there (force_coop 0 -> K = 4 up to ~305,000 bodies).  test_force_coop_walk_matches_the_one_wave_walk
    compares that walk with the one-wave walk; this compares it with the ORACLE directly, same stated tolerance as
    test_force_fast_vs_oracle: relative |da| median <= 2e-6, 99.9th percentile <= 2e-5, max <= 5e-4."""
    ic = pkg.plummer(n, seed=42)
    e = _engine(pkg, ic)
    assert e.params.force_coop == 0
    e.tree_stages(); e.force()
    ga = np.stack(e.download_acc(), 1)
    p = oparams(orc, e.params)
    o = oracle_pipeline(orc, ic, p)
    oacc, *_ = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    oa = np.zeros((n, 3), np.float32)
    oa[o["perm"]] = oacc[:, :3]
    rel = np.linalg.norm(ga - oa, axis=1) / np.linalg.norm(oa, axis=1)
    print(f"default engine (K = 4 everywhere) vs oracle, n={n}: p50 {np.median(rel):.2e} p99.9 {np.quantile(rel, 0.999):.2e} "
          f"max {rel.max():.2e}")
    assert np.median(rel) <= 2e-6 and np.quantile(rel, 0.999) <= 2e-5 and rel.max() <= 5e-4
    assert e.stats().status_flags == 0 and e.stats().force_redo_waves == 0
    e.close()


def test_force_walk_stats_match_the_oracles_group_walk(pkg, orc):
    """bh_force_walk_stats (the counters bench.py prices the issue-rate roofline with): blocks popped by the
    64-body waves == pops of the oracle's group walk on the same tree (MAC ties aside), pairs = the records
    of those blocks two at a time, and the in-kernel clock is a plausible shader clock."""
    n = 65536
    ic = pkg.plummer(n, seed=42)
    e = _engine(pkg, ic)
    e.tree_stages()
    ws = e.force_walk_stats()
    rec = e.download_tree()
    bodies = e.download_sorted_bodies()
    p = oparams(orc, e.params)
    g = orc.group_stats(rec, bodies, p, 64, 1)
    assert ws.waves == g["groups"] == n // 64
    assert abs(int(ws.blocks) - g["pops"]) <= 2e-3 * g["pops"]
    assert 2 * ws.pairs >= g["records"] * (1 - 2e-3) and 2 * ws.pairs <= (g["records"] + g["pops"]) * (1 + 2e-3)
    assert 0 < ws.masked_pairs < ws.pairs
    assert 1.0 < ws.clock_ghz < 2.6 and ws.wave_cycles_max >= ws.wave_cycles_mean > 0
    e.close()


def test_force_theta0_is_direct_sum(pkg, orc):
    """theta = 0 opens every cell: the traversal must touch every body exactly once."""
    n = 2048
    ic = pkg.plummer(n, seed=8)
    e = _engine(pkg, ic, theta=0.0)
    e.tree_stages(); e.force_count()
    gV, gO, gP = e.download_counters()
    assert np.all(gP == n)
    ax, ay, az = e.download_acc()
    xyzm = np.stack([ic[0], ic[1], ic[2], ic[6]], 1)
    d = orc.direct_f64(xyzm, e.params.G, e.params.eps2)
    rel = np.linalg.norm(np.stack([ax, ay, az], 1) - d, axis=1) / np.linalg.norm(d, axis=1)
    assert rel.max() <= 2e-5  # fp32 accumulation of 2048 terms
    e.close()


@pytest.mark.parametrize("n", [1, 1000, 65536])
def test_integrate_bit_exact(pkg, orc, n):
    ic = pkg.plummer(n, seed=6)
    e = _engine(pkg, ic, max_speed=0.75)  # low clamp so the speed-limit branch is exercised
    e.tree_stages(); e.force()
    bodies = e.download_sorted_bodies()
    order = e.download_order()
    ax, ay, az = e.download_acc()
    acc4 = np.zeros((n, 4), np.float32)
    acc4[:, 0] = ax[order]; acc4[:, 1] = ay[order]; acc4[:, 2] = az[order]
    vel = np.stack([ic[3], ic[4], ic[5]], 1)[order]
    p = oparams(orc, e.params)
    oxyzm, ovel = orc.integrate(bodies, vel, acc4, p)
    speed = np.linalg.norm(ovel, axis=1)
    if n >= 1000:
        assert (speed >= 0.75 * 0.999).any() and (speed < 0.7).any(), "clamp branch not exercised"
    e.integrate()
    x, y, z, vx, vy, vz = e.download()
    assert np.array_equal(x[order], oxyzm[:, 0])
    assert np.array_equal(y[order], oxyzm[:, 1])
    assert np.array_equal(z[order], oxyzm[:, 2])
    assert np.array_equal(vx[order], ovel[:, 0])
    assert np.array_equal(vy[order], ovel[:, 1])
    assert np.array_equal(vz[order], ovel[:, 2])
    e.close()


@pytest.mark.parametrize("n,steps", [(4096, 100), (65536, 10)])
def test_steps_end_to_end(pkg, orc, n, steps):
    """K whole steps, Plummer sphere, theta = 0.5 (BASELINE config 1 at 65,536) vs the oracle's step loop, as a
    distribution over the bodies of max(|dx|,|dy|,|dz|) and max |dv| (positions: sphere scale a = 400, one ulp of
    a coordinate below 512 is 3.1e-5; velocities ~10).  Stated tolerance (strict kernel / fast kernel): see TOL_STEPS
    below (round-3 review: the bounds were 1e-2 / 2e-3, two orders of magnitude above what is measured)."""
    ic = pkg.plummer(n, seed=42)
    o = orc.Oracle(n)
    o.upload(*ic)
    o.step(steps, order=orc.ORDER_BATCHED)
    ref = np.stack(o.download(), 1).astype(np.float64)
    for strict in (1, 0):
        e = _engine(pkg, ic, strict_fp=strict)
        e.step(steps)
        got = np.stack(e.download(), 1).astype(np.float64)
        st = e.stats()
        assert st.status_flags == 0 and st.steps == steps
        dx = np.abs(got[:, :3] - ref[:, :3]).max(axis=1)
        dv = np.abs(got[:, 3:] - ref[:, 3:]).max(axis=1)
        print(f"n={n} K={steps} strict={strict}: |dx| p50 {np.median(dx):.3e} p99.9 {np.percentile(dx, 99.9):.3e} max {dx.max():.3e}; "
              f"|dv| p50 {np.median(dv):.3e} p99.9 {np.percentile(dv, 99.9):.3e} max {dv.max():.3e}")
        t = TOL_STEPS[(n, strict)]
        assert np.median(dx) <= t[0] and np.percentile(dx, 99.9) <= t[1] and dx.max() <= t[2], (strict, t)
        assert np.median(dv) <= t[3] and np.percentile(dv, 99.9) <= t[4] and dv.max() <= t[5], (strict, t)
        e.close()
    o.close()


# (n, strict) -> |dx| median, p99.9, max, |dv| median, p99.9, max.  Measured on MI355X in round 4
# (profiles/r04_parity/steps_end_to_end.txt; positions reach ~500, where one ulp is 3.05e-5, velocities ~10):
#   4,096 x 100 strict  |dx| 0 / 3.6e-6 / 3.1e-5   |dv| 0 / 2.4e-7 / 4.8e-7      fast  0 / 1.5e-5 / 3.1e-5   0 / 2.4e-7 / 4.8e-7
#   65,536 x 10 strict  |dx| 0 / 0 / 1.5e-5        |dv| 0 / 9.5e-7 / 1.9e-6      fast  0 / 0 / 3.1e-5        0 / 9.5e-7 / 2.4e-6
# bounds: one ulp of a position at the median and the 99.9th percentile, two at the maximum; velocities 2x measured
TOL_STEPS = {(4096, 1): (3.1e-5, 3.1e-5, 6.2e-5, 1e-7, 5e-7, 1e-6), (4096, 0): (3.1e-5, 3.1e-5, 6.2e-5, 1e-7, 5e-7, 1e-6),
             (65536, 1): (3.1e-5, 3.1e-5, 3.1e-5, 1e-7, 2e-6, 4e-6), (65536, 0): (3.1e-5, 3.1e-5, 6.2e-5, 1e-7, 2e-6, 5e-6)}


def _state(e):
    return np.stack(e.download(), 1).tobytes(), e.download_order().tobytes()


@pytest.mark.parametrize("n", [1, 2, 63, 65, 1000, 24577, 81921, 250000])
def test_fused_force_integrate_step_matches_separate_kernels(pkg, n):
    """bh_step's force launch also integrates and folds the next cube (modes other than per-stage timing); under
    bh_set_timing(1) the same step runs force and integrate as separate kernels.  Same states, same order and the
    same cube after 4 steps, for one wave, the 16 / 32 / 64 bodies-per-wave launches and both walk instances"""
    ic = pkg.plummer(n, seed=31)
    out = []
    for timing in (0, 1):
        e = _engine(pkg, ic)
        e.set_timing(timing)
        e.step(4)
        assert e.stats().status_flags == 0
        out.append((_state(e), e.download_bounds().tobytes()))
        e.close()
    assert out[0] == out[1]


def test_step_shortcuts_survive_uploads_and_stage_calls(pkg):
    """bh_step takes two shortcuts from the previous step — the bounding cube folded by its integrate kernel and
    the splitter sort that relies on the stored key order — and both must be dropped whenever something else
    wrote the bodies.  Every mixed sequence below must end bit-identical to the same physics on a context that
    never takes a shortcut (radix sort pinned, one stage call at a time)."""
    n = 50000
    ic1 = pkg.plummer(n, seed=1)
    ic2 = special_ics("outlier", n, np.random.default_rng(2))

    def stages(e, k):
        for _ in range(k):
            e.bbox(); e.morton(); e.sort(); e.build(); e.com(); e.force(); e.integrate()

    ref = _engine(pkg, ic1, sort_variant=2)
    stages(ref, 3)
    ref.upload(*ic2)
    stages(ref, 4)
    want = _state(ref)
    ref.close()

    # (a) steps, upload in the middle (different cube, caller order), steps
    e = _engine(pkg, ic1)
    e.step(3)
    e.upload(*ic2)
    e.step(4)
    assert _state(e) == want
    assert e.stats().status_flags == 0
    e.close()
    # (b) steps and single-stage calls interleaved; morton twice before a sort; a bbox the step does not need
    e = _engine(pkg, ic1)
    e.step(1)
    stages(e, 1)
    e.bbox(); e.morton(); e.bbox(); e.morton(); e.sort(); e.build(); e.com(); e.force(); e.integrate()
    e.upload(*ic2)
    stages(e, 1)
    e.step(2)
    e.bbox()
    e.step(1)
    assert _state(e) == want
    assert e.stats().status_flags == 0
    e.close()


def test_step_graph_replay_matches_plain_launches(pkg):
    """bh_params.step_graph = 1: bh_step captured once per body-array parity and replayed as a HIP graph (radix
    sort and explicit bbox kernels: every kernel argument of a step is then constant) == plain launches, bit for
    bit, over 7 steps and across an upload"""
    n = 30000
    ic1 = pkg.plummer(n, seed=4)
    ic2 = pkg.plummer(n, seed=5)
    out = []
    for g in (0, 1):
        e = _engine(pkg, ic1, step_graph=g)
        e.step(4)
        e.upload(*ic2)
        e.step(3)
        out.append(_state(e))
        assert e.stats().status_flags == 0
        e.close()
    assert out[0] == out[1]


def test_timing_modes(pkg):
    """bh_set_timing: 1 = an event after every stage (every ms_* of bh_stats filled, the history carries force and
    step times), 2 = only the pair around the force launch (what bench.py's timed region uses: ms_force and the
    force history only), 3 = that pair on every 4th step (steps 0 and 4 of these 5), 0 = off; the physics does not
    depend on it — mode 1 steps with separate force and integrate kernels, the others with the fused launch"""
    n = 40000
    ic = pkg.plummer(n, seed=12)
    states = []
    for mode in (0, 1, 2, 3):
        e = _engine(pkg, ic)
        e.set_timing(mode)
        e.step(5)
        st = e.stats()
        f, t = e.timing_history()
        if mode == 0:
            assert len(f) == 0 and st.ms_force == 0.0
        elif mode == 1:
            assert len(f) == 5 and (f > 0).all() and (t > f).all()
            assert min(st.ms_morton, st.ms_sort, st.ms_build, st.ms_com, st.ms_force, st.ms_integrate) > 0.0
            assert st.ms_step >= st.ms_force
        else:
            assert len(f) == (5 if mode == 2 else 2) and (f > 0).all() and (t == 0).all()
            assert st.ms_force > 0.0 and st.ms_sort == 0.0 and st.ms_step == 0.0
        states.append(_state(e))
        e.close()
    assert states[0] == states[1] == states[2] == states[3]


@pytest.mark.parametrize("n", [5000, 70001, 300000])
def test_step_cube_from_integrate_is_the_bbox_cube(pkg, orc, n):
    """the cube a step takes from the previous step — folded by the force launch that also integrates (16 / 32 / 64
    bodies per wave at these sizes: 313 / 2,188 / 4,688 waves, two-level hand-off) — == bh_bbox of the same positions"""
    ic = pkg.plummer(n, seed=9)
    e = _engine(pkg, ic)
    for _ in range(3):
        e.step(1)
        x, y, z = e.download()[:3]
        want = orc.bbox(x, y, z)
        e.step(1)                      # uses the folded cube; the tree it leaves was built inside it
        used = e.download_bounds()
        assert np.array_equal(used, want)
    e.close()


def test_stage_order_errors(pkg):
    ic = pkg.plummer(100, seed=1)
    e = pkg.Engine(100)
    with pytest.raises(pkg.BhError):
        e.step()            # nothing uploaded
    e.upload(*ic)
    with pytest.raises(pkg.BhError):
        e.morton()          # bbox first
    e.bbox(); e.morton(); e.sort()
    with pytest.raises(pkg.BhError):
        e.sort()            # sorting twice would permute twice
    with pytest.raises(pkg.BhError):
        e.force()           # build/com first
    e.build(); e.com(); e.force(); e.integrate()
    with pytest.raises(pkg.BhError):
        e.integrate()       # force first
    e.close()
    with pytest.raises(pkg.BhError):
        pkg.Engine(0)
    with pytest.raises(pkg.BhError):
        pkg.Engine(10, eps2=0.0)
    with pytest.raises(pkg.BhError):
        pkg.Engine(10, key_bits=48)


@pytest.mark.parametrize("leaf_cap", [1, 4])
def test_canonical_tree_records_exist_when_asked_for(pkg, orc, leaf_cap):
    """bh_step of the default engine writes only the force kernel's digests (the COM stage skips the canonical
    x/y/z/m and body-range arrays: 24 B less traffic per record).  The reference's d_nodes is readable after every
    simulationStep (ref:266-281), so bh_download_tree produces the canonical records on demand (canon_kernel: the
    step's prefix sums and digests are still there; the bodies have moved since, so a BODY record's position comes
    from its digest).  Checked against (a) a second engine that reaches the same tree through the stage calls —
    bit for bit, every field — and (b) the oracle's tree of that state.  After bh_build alone (no centre of mass
    yet) the download still refuses."""
    n = 3000
    ic = pkg.plummer(n, seed=17)
    e = _engine(pkg, ic, leaf_cap=leaf_cap)
    e.step(2)
    rec_a = e.download_tree()       # tree of step 2, made canonical now
    rec_a2 = e.download_tree()      # asking twice changes nothing
    assert rec_a.tobytes() == rec_a2.tobytes()
    assert e.stats().status_flags == 0
    b = _engine(pkg, ic, leaf_cap=leaf_cap)
    b.step(1)
    state = [np.ascontiguousarray(a) for a in b.download()]
    mass = b.download_mass()
    b.bbox(); b.morton(); b.sort(); b.build()
    with pytest.raises(pkg.BhError):
        b.download_tree()           # centres of mass not set yet
    b.com()
    rec_b = b.download_tree()
    assert len(rec_a) == len(rec_b)
    for f in rec_b.dtype.names:
        assert rec_a[f].tobytes() == rec_b[f].tobytes(), f
    p = oparams(orc, b.params)
    o = oracle_pipeline(orc, tuple(state) + (mass,), p)
    for f in ("kind", "first", "count"):
        assert np.array_equal(rec_a[f], o["rec"][f]), f
    live = rec_a["kind"] != 3
    scale = float(np.abs(o["rec"]["x"][live]).max())
    for f in ("x", "y", "z"):
        assert np.abs(rec_a[f][live] - o["rec"][f][live]).max() <= 4 * np.spacing(np.float32(scale)), f
    assert np.allclose(rec_a["m"][live], o["rec"]["m"][live], rtol=1e-6)
    # the step after an on-demand download is unaffected by it
    e.step(1); b.force(); b.integrate(); b.step(1)
    assert all(np.array_equal(u, v) for u, v in zip(e.download(), b.download()))
    # the tree of a step survives the stage calls of the next one up to the sort (records made canonical in time)
    e.bbox(); e.morton(); e.sort()
    rec_c = e.download_tree()
    assert (rec_c["kind"] == 1).sum() > 0 and np.isfinite(rec_c["x"]).all()
    assert e.stats().status_flags == 0 and b.stats().status_flags == 0
    e.close(); b.close()
    s = _engine(pkg, ic, strict_fp=1)
    s.step(2)
    rec2 = s.download_tree()       # a strict step reads the canonical records, so it writes them
    assert (rec2["kind"] == 1).sum() > 0 and np.isfinite(rec2["x"]).all()
    s.close()


def test_tree_download_after_replayed_graph_steps(pkg):
    """bh_params.step_graph = 1: the third and fourth bh_step REPLAY the graphs the first two captured — the host-side
    bookkeeping of the COM stage (records are proto again, digests belong to this tree) must advance as it does next
    to the real launches, or bh_download_tree after a replayed step skips canon_kernel and returns the body-range bit
    patterns (round-4 advisor finding).  Tree after every step == the tree of an engine that launches its steps."""
    n = 5000
    ic = pkg.plummer(n, seed=23)
    g = _engine(pkg, ic, step_graph=1)
    p = _engine(pkg, ic)
    for k in range(5):
        g.step(1)
        p.step(1)
        ra, rb = g.download_tree(), p.download_tree()
        assert len(ra) == len(rb), k
        for f in rb.dtype.names:
            assert ra[f].tobytes() == rb[f].tobytes(), (k, f)
        assert np.isfinite(ra["x"]).all()
    assert all(np.array_equal(u, v) for u, v in zip(g.download(), p.download()))
    # and through the stage calls after a replayed step (bh_sort makes the last step's records canonical in time)
    g.bbox(); g.morton(); g.sort()
    p.bbox(); p.morton(); p.sort()
    assert g.download_tree().tobytes() == p.download_tree().tobytes()
    assert g.stats().status_flags == 0
    g.close(); p.close()


@pytest.mark.parametrize("leaf_cap", [1, 4])
def test_com_stage_is_reentrant(pkg, leaf_cap):
    """bh_com twice on one tree (round-3 review: the second call took a record's body range from the bit patterns of
    x / y, which the first call had replaced by the centre of mass): same records, same accelerations, no flags."""
    n = 5000
    ic = pkg.plummer(n, seed=23)
    e = _engine(pkg, ic, leaf_cap=leaf_cap)
    e.tree_stages()
    rec1 = e.download_tree()
    e.force()
    a1 = e.download_acc()
    e.com(); e.com()
    rec2 = e.download_tree()
    e.force()
    a2 = e.download_acc()
    assert rec1.tobytes() == rec2.tobytes()
    assert all(np.array_equal(u, v) for u, v in zip(a1, a2))
    assert e.stats().status_flags == 0
    e.close()


@pytest.mark.parametrize("coop,n", [(1, 10000), (0, 10000), (0, 300000)])
def test_force_range_matches_full(pkg, coop, n):
    """bh_force_range (the multi-rank shard entry point) == the same rows of a full bh_force.
    The one-wave-per-group walk (force_coop = 1) sums a body's interactions in an order that does not depend on
    which other bodies share its wave, so ANY split is bit-identical.  With several waves per group (the default up
    to ~305,000 bodies, and the last groups of every larger launch) the order depends on the group's composition:
    bit-identical for splits on 64-body group boundaries (what dist.ShardedStepper uses: 256-aligned slabs); a slab
    that starts inside a group of that part is REFUSED (BH_ERR_BAD_ARG, round-4 advisor finding: it used to return
    bits that differ from the full launch's).  Which walk a group gets is decided from the context's body count,
    never from the range, so a slab launch cannot change it."""
    ic = pkg.plummer(n, seed=12)
    e = _engine(pkg, ic, force_coop=coop)
    e.tree_stages(); e.force()
    full = np.stack(e.download_acc(), 1)
    q = n // 4 // 256 * 256
    for ranges in (((0, q), (q, q + 64), (q + 64, 3 * q), (3 * q, n)),
                   ((0, q - 60), (q - 60, q - 59), (q - 59, 3 * q + 17), (3 * q + 17, n))):
        e2 = _engine(pkg, ic, force_coop=coop)
        e2.tree_stages()
        aligned = all(lo % 64 == 0 for lo, _ in ranges)
        if coop == 1 or aligned:
            for lo, hi in ranges:
                e2.force(lo, hi)
            part = np.stack(e2.download_acc(), 1)
            assert np.array_equal(full, part), (coop, n, ranges)
        else:   # (both sizes are walked cooperatively throughout: every group boundary is a multiple of 64)
            for lo, hi in ranges:
                if lo % 64 == 0:
                    e2.force(lo, hi)
                else:
                    with pytest.raises(pkg.BhError) as ei:
                        e2.force(lo, hi)
                    assert ei.value.status == -1
        assert e2.stats().status_flags == 0
        e2.close()
    e.close()


def _nested_corner_ic(levels=18, S=float(2 ** 22)):
    """Cells nested towards the MAX corner of the cube, seven two-body sibling cells per level: the
    nested cell is the last child of every block, so a depth-first walk of the deepest bodies keeps seven
    opened siblings pending per level — a traversal stack ~7 x levels deep."""
    pts = []
    for L in range(1, levels + 1):
        h = S * 2.0 ** -L
        for o in range(1, 8):                       # the seven octants that are not the nested one
            c = np.array([(o >> 2) & 1, (o >> 1) & 1, o & 1], np.float64) * h + h / 2
            pts.append(c + np.array([h / 4, 0, 0]))
            pts.append(c - np.array([h / 4, 0, 0]))
    h = S * 2.0 ** -(levels + 1)
    for o in range(8):                              # the innermost cell: eight bodies
        pts.append(np.array([(o >> 2) & 1, (o >> 1) & 1, o & 1], np.float64) * h + h / 2)
    u = np.array(pts)
    p = (S - u).astype(np.float32)                  # nested corner = max corner: octant 7 at every level
    n = len(p)
    z = np.zeros(n, np.float32)
    return p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy(), z, z.copy(), z.copy(), np.full(n, 3.0, np.float32)


def test_force_fast_deep_stack_falls_back_to_the_large_stack(pkg, orc):
    """the fast kernel walks with a 64-entry stack and redoes a wave with the 192-entry one when that
    overflows: ~7 x 18 pending entries here; the result must still match the oracle and no flag is set"""
    ic = _nested_corner_ic()
    n = len(ic[0])
    e = _engine(pkg, ic, theta=0.3)   # every sibling fails s/d < 0.3 for the innermost bodies: all 7 opened
    e.tree_stages()
    e.force()
    ax, ay, az = e.download_acc()
    st = e.stats()
    assert st.status_flags == 0 and st.max_level >= 18
    p = oparams(orc, e.params)
    o = oracle_pipeline(orc, ic, p)
    oacc, *_ = orc.force(o["rec"], o["xyzm"], p, orc.ORDER_PREORDER)
    oa = np.zeros((n, 3), np.float32)
    oa[o["perm"]] = oacc[:, :3]
    ga = np.stack([ax, ay, az], 1)
    rel = np.linalg.norm(ga - oa, axis=1) / np.maximum(np.linalg.norm(oa, axis=1), 1e-30)
    assert np.median(rel) <= 2e-6 and rel.max() <= 2e-4, (np.median(rel), rel.max())
    # the counting kernel (3 x 64 entries) sees the same tree: every body opens at least one cell per level
    e.force_count()
    V, O, P = e.download_counters()
    assert O.max() >= 7 * 17      # >= 119 opened cells on one body's walk, seven pending per level
    e.close()
