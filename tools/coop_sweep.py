"""Force-stage time of the cooperative walk (bh_params.force_coop = K waves per group) against the one-wave walk,
per body count, group size and K: same box, same tree, 30 launches each.   python tools/coop_sweep.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402

pkg = bhpkg.load()
sizes = [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536, 125000, 200000, 300000, 500000, 1000000]


def force_ms(n, ic, reps=30, **kw):
    e = pkg.Engine(n, **kw)
    e.upload(*ic)
    e.step(3)
    e.tree_stages()
    for _ in range(3):
        e.force()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.force()
    e.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    redo = e.stats().force_redo_waves
    e.close()
    return ms, redo


for n in sizes:
    ic = pkg.plummer(n, seed=42)
    base = {g: force_ms(n, ic, force_coop=1, force_group=g)[0] for g in (16, 32, 64)}
    print(f"n={n}: one wave per group  16: {base[16]:.4f}  32: {base[32]:.4f}  64: {base[64]:.4f} ms", flush=True)
    for g in (32, 64):
        row = []
        for K in (2, 3, 4, 5, 6, 8):
            groups = (n + g - 1) // g
            if groups * K > 40000 and K > 4:
                continue
            ms, redo = force_ms(n, ic, force_coop=K, force_group=g)
            row.append(f"K={K}: {ms:.4f}" + (f" (redo {redo})" if redo else ""))
        print(f"   group {g}: " + "  ".join(row), flush=True)
