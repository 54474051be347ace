"""Fuzz of the cooperative force walk against the one-wave walk: random body counts, theta, leaf_cap, depth caps,
initial conditions (Plummer, disc, clumps with coincident bodies), waves per group and group sizes; the accelerations
must agree to the association of fp32 sums (relative |da| median <= 3e-6, max <= 5e-4 of a body's |a|; absolute for
bodies whose force nearly cancels), no device flag may be raised, and two runs of the same setting must give the same
bits.   python tools/coop_fuzz.py [cases] [seed] [log10 of the largest n: 5.55; 6.2 reaches the mixed launch]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402

pkg = bhpkg.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
top = float(sys.argv[3]) if len(sys.argv) > 3 else 5.55
bad = 0
for t in range(cases):
    n = int(10 ** rng.uniform(0.3 if top < 6 else 5.3, top))
    theta = float(rng.choice([0.2, 0.3, 0.5, 0.5, 0.8, 1.0]))
    leaf_cap = int(rng.choice([1, 1, 1, 4, 8, 16]))
    max_depth = int(rng.choice([21, 21, 21, 8, 5]))
    kind = str(rng.choice(["plummer", "plummer", "disc", "clumps"]))
    K = int(rng.choice([0, 2, 3, 4, 8]))
    group = int(rng.choice([0, 16, 32, 64]))
    seed = int(rng.integers(1, 1 << 30))
    if kind == "disc":
        ic = [a.copy() for a in pkg.disc(n, seed=seed)]
    else:
        ic = [a.copy() for a in pkg.plummer(n, seed=seed)]
    if kind == "clumps" and n > 50:
        for _ in range(int(rng.integers(1, 6))):
            lo = int(rng.integers(0, n - 20))
            cnt = int(rng.integers(2, min(40, n - lo)))
            for a in ic[:3]:
                a[lo:lo + cnt] = a[lo]
        ic[6][: n // 7] = 0.0
    ic = tuple(ic)
    kw = dict(theta=theta, leaf_cap=leaf_cap, max_depth=max_depth)

    def acc(**more):
        e = pkg.Engine(n, **kw, **more)
        e.upload(*ic)
        e.tree_stages()
        e.force()
        a = np.stack(e.download_acc(), 1)
        st = e.stats()
        e.close()
        return a, st
    ref, st0 = acc(force_coop=1)
    a1, st1 = acc(force_coop=K, force_group=group)
    a2, st2 = acc(force_coop=K, force_group=group)
    norm = np.sqrt((ref.astype(np.float64) ** 2).sum(1))
    scale = max(float(norm.max()), 1e-30)
    err = np.sqrt(((a1.astype(np.float64) - ref) ** 2).sum(1))
    rel = err / np.maximum(norm, 1e-3 * scale)
    ok = (st0.status_flags == 0 and st1.status_flags == 0 and a1.tobytes() == a2.tobytes() and np.isfinite(a1).all()
          and np.median(rel) <= 3e-6 and rel.max() <= 5e-4)
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} case {t}: n {n} theta {theta} leaf_cap {leaf_cap} depth {max_depth} {kind} K {K} group {group}: "
          f"rel median {np.median(rel):.2e} max {rel.max():.2e} redo {st1.force_redo_waves} flags {st1.status_flags}", flush=True)
print(f"{cases - bad} ok, {bad} failed")
sys.exit(1 if bad else 0)
