"""force time of the stage call bh_force (not fused with the integrate step) on the tree of a fixed state: wall clock over
back-to-back launches.  python tools/force_stage_ms.py [n] [theta] [reps]   (BH_LIB_PATH / BH_FORCE_* select study builds)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
e = pkg.Engine(n, theta=theta, force_coop=int(os.environ.get("BH_COOP", "0")))
e.upload(*pkg.plummer(n, seed=42))
e.step(3)
e.tree_stages()
for _ in range(3):
    e.force()
e.sync()
best = []
for r in range(3):
    t0 = time.perf_counter()
    for _ in range(reps):
        e.force()
    e.sync()
    best.append((time.perf_counter() - t0) / reps * 1e3)
a = np.stack(e.download_acc(), 1)
print(f"n={n} theta={theta} graded={os.environ.get('BH_FORCE_GRADED', '0')}: force {min(best):.4f} ms (runs {[round(b, 4) for b in best]}) checksum {float(np.abs(a).sum()):.6e} flags {e.stats().status_flags}")
