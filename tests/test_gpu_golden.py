"""GPU parity against the COMMITTED fixtures tests/golden/plummer4096_seed42.* — no oracle library is imported, built
or loaded here, so these checks do not depend on anything compiled on the GPU box except libbh.so itself.  The
fixtures were written by tests/golden/make_golden.py (this repo's CPU oracle with its default parameters: Morton key
order, theta 0.5, leaf_cap 1, 63-bit keys); the CPU suite (test_oracle.py::test_golden_fixtures) pins the oracle to
the same file, so the two sides meet in the data.  Reference lines: nbody_v5_bench.cu:134-156 (cube), :42-63 (30-bit
keys), :262-264 (sort), :83-132 (tree), :158-189 (COM), :191-225 (force), :227-249 (integrate), :255-283 (step)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    g = np.load(os.path.join(GOLD, "plummer4096_seed42.npz"))
    meta = json.load(open(os.path.join(GOLD, "plummer4096_seed42.json")))
    return g, meta


def _ic(g):
    return tuple(g[k] for k in ("x", "y", "z", "vx", "vy", "vz", "m"))


def test_golden_ic_generator(pkg, gold):
    g, meta = gold
    ic = pkg.plummer(meta["n"], seed=meta["seed"])
    for a, b, name in zip(ic, _ic(g), "x y z vx vy vz m".split()):
        assert np.array_equal(a, b), name


def test_golden_cube_keys_order(pkg, gold):
    """bbox (ref:134-156), 63-bit Morton keys + stable sort (ref:262-264) and the reference-literal 30-bit Morton
    code (ref:42-63): bit for bit"""
    g, meta = gold
    n = meta["n"]
    e = pkg.Engine(n, key_curve=0)
    e.upload(*_ic(g))
    e.bbox()
    assert e.download_bounds().tobytes() == g["bounds"].tobytes()
    e.morton(); e.sort()
    assert np.array_equal(e.download_keys(), g["sorted_keys"])
    assert np.array_equal(e.download_order(), g["perm"])
    assert e.stats().status_flags == 0
    e.close()
    e = pkg.Engine(n, key_bits=30, max_depth=10)
    e.upload(*_ic(g))
    e.bbox(); e.morton()
    assert np.array_equal(e.download_keys().astype(np.uint32), g["morton30"])
    e.close()


@pytest.mark.parametrize("via_step", [False, True])
def test_golden_tree_records(pkg, gold, via_step):
    """topology and cell edges bit for bit; centres of mass within 4 ulp of the coordinate scale, masses to 1e-6
    relative (fp64 prefix differences here, fp64 per-cell sums in the fixture).  via_step: the records bh_download_tree
    produces on demand after a whole bh_step (whose COM stage wrote only digests) are those of the step's tree, i.e.
    of the uploaded state."""
    g, meta = gold
    n = meta["n"]
    e = pkg.Engine(n, key_curve=0)
    e.upload(*_ic(g))
    if via_step:
        e.step(1)
    else:
        e.tree_stages()
    rec = e.download_tree()
    st = e.stats()
    assert st.status_flags == 0
    assert st.n_internal == meta["n_internal"] and len(rec) == meta["n_entries"] and st.max_level == meta["max_level"]
    for f in ("kind", "first", "count"):
        assert np.array_equal(rec[f], g["rec_" + f]), f
    assert rec["s"].tobytes() == g["rec_s"].tobytes()
    live = rec["kind"] != 3
    scale = float(np.abs(g["rec_x"][live]).max())
    for f in ("x", "y", "z"):
        assert np.abs(rec[f][live] - g["rec_" + f][live]).max() <= 4 * np.spacing(np.float32(scale)), f
    assert np.allclose(rec["m"][live], g["rec_m"][live], rtol=1e-6, atol=0)
    e.close()


def test_golden_strict_force_and_counters(pkg, gold):
    """strict_fp walk (the reference's source-text arithmetic, ref:203-213): V / O / P per body equal the fixture's
    exactly (same accept / open decisions for every body); accelerations within 2e-6 of the largest |a| (the
    centres of mass differ by ulps, the summation order is the same pre-order)"""
    g, meta = gold
    n = meta["n"]
    e = pkg.Engine(n, key_curve=0, strict_fp=1)
    e.upload(*_ic(g))
    e.tree_stages()
    e.force_count()
    V, O, P = e.download_counters()
    perm = g["perm"]
    assert np.array_equal(V[perm], g["V"]) and np.array_equal(O[perm], g["O"]) and np.array_equal(P[perm], g["P"])
    a = np.stack(e.download_acc(), 1)[perm]
    ref = g["acc_preorder"][:, :3]
    amax = float(np.sqrt((ref.astype(np.float64) ** 2).sum(1)).max())
    err = np.abs(a - ref).max()
    print(f"strict |da| max {err:.3e}, |a| max {amax:.3e}")
    assert err <= 2e-6 * amax
    assert e.stats().status_flags == 0
    e.close()


def test_golden_fast_force(pkg, gold):
    """the benchmarked walk (packed fma, v_rsq_f32): relative |da| per body against the fixture — median <= 1.2e-6,
    max <= 1e-4 (the bounds DESIGN.md §2 states for the full-size runs, which this small case stays well inside)"""
    g, meta = gold
    n = meta["n"]
    e = pkg.Engine(n, key_curve=0)
    e.upload(*_ic(g))
    e.tree_stages()
    e.force()
    a = np.stack(e.download_acc(), 1)[g["perm"]].astype(np.float64)
    ref = g["acc_preorder"][:, :3].astype(np.float64)
    rel = np.sqrt(((a - ref) ** 2).sum(1)) / np.sqrt((ref ** 2).sum(1))
    print(f"fast rel |da| median {np.median(rel):.3e} p99.9 {np.percentile(rel, 99.9):.3e} max {rel.max():.3e}")
    assert np.median(rel) <= 1.2e-6 and rel.max() <= 1e-4
    assert e.stats().status_flags == 0
    e.close()


def test_golden_integrate_and_ten_steps(pkg, gold):
    """one integrate on the fixture's accelerations is covered bit for bit by the oracle-backed tests; here: the
    state after 10 whole steps (ref:255-283) against the fixture's, caller order.  Strict engine (same arithmetic
    as the fixture up to the ulp-level COM differences) and the fast engine; bounds: TOL10_* below."""
    g, meta = gold
    n = meta["n"]
    ref = g["state_after_10"]
    for strict, tol in ((1, TOL10_STRICT), (0, TOL10_FAST)):
        e = pkg.Engine(n, key_curve=0, strict_fp=strict)
        e.upload(*_ic(g))
        e.step(10)
        got = np.stack(e.download(), 1)
        assert e.stats().status_flags == 0
        dx = np.abs(got[:, :3] - ref[:, :3]).max(axis=1)
        dv = np.abs(got[:, 3:] - ref[:, 3:]).max(axis=1)
        print(f"golden K=10 strict={strict}: |dx| p50 {np.median(dx):.3e} p99.9 {np.percentile(dx, 99.9):.3e} "
              f"max {dx.max():.3e}; |dv| p50 {np.median(dv):.3e} p99.9 {np.percentile(dv, 99.9):.3e} max {dv.max():.3e}")
        assert np.median(dx) <= tol[0] and dx.max() <= tol[1], (strict, tol)
        assert np.median(dv) <= tol[2] and dv.max() <= tol[3], (strict, tol)
        e.close()


# |dx| median, max, |dv| median, max.  Measured in round 4 (profiles/r04_parity/golden.txt): strict 0 / 0 / 0 / 0 — the
# same bits as the fixture for every body —, fast 0 / 0 / 0 / 2.4e-7.  Bounds: one ulp of a position (3.05e-5 below
# 512), 2x the measured velocity deviation (strict: one ulp of a velocity ~10).
TOL10_STRICT = (0.0, 3.1e-5, 0.0, 1e-6)
TOL10_FAST = (0.0, 3.1e-5, 1e-7, 5e-7)
