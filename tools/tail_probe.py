"""Is the 1M-body force launch tail-bound?  Two independent contexts (own streams) run their force
stage (a) one after the other, (b) concurrently.  If (b) is clearly cheaper than (a), the tail of one
launch is being filled by the other."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
engs = []
for k in range(K):
    e = pkg.Engine(n)
    e.upload(*pkg.plummer(n, seed=42 + k))
    e.tree_stages()
    e.force(); e.sync()
    engs.append(e)
def run(conc, reps=10):
    best = 1e9
    for _ in range(reps):
        for e in engs: e.sync()
        t0 = time.perf_counter()
        if conc:
            for e in engs: e.force()
            for e in engs: e.sync()
        else:
            for e in engs:
                e.force(); e.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"n={n} K={K}: sequential {run(False):.3f} ms, concurrent {run(True):.3f} ms")
