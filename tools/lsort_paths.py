"""design study: which path every bucket of local_sort_kernel took on a clustered input (trace build, BH_LIB_PATH):
passes per bucket = 3-4 (top window only, runs ordered by exchanges) or 4 + the full set (a run too long)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = 30000
rng = np.random.default_rng(17)
x = rng.uniform(-1000.0, 1000.0, (n, 3))
k = 1000
for _ in range(300):
    x[k + 1] = x[k] + rng.normal(0, 1e-5, 3); k += 2
for _ in range(40):
    m = int(rng.integers(5, 31)); x[k:k + m] = x[k] + rng.normal(0, 1e-5, (m, 3)); k += m
for wdt in (0.5, 0.1, 0.02, 1e-4):
    x[k:k + 700] = x[k] + rng.uniform(-wdt, wdt, (700, 3)); k += 700
x[k:k + 50] = x[k]; k += 50
cc = rng.uniform(-900.0, 900.0, (64, 3))
x[k:k + 9600] = cc[np.arange(9600) % 64] + rng.uniform(-1e-3, 1e-3, (9600, 3))
x = x.astype(np.float32)
z = np.zeros(n, np.float32)
e = pkg.Engine(n, sort_variant=3)
e.upload(x[:, 0].copy(), x[:, 1].copy(), x[:, 2].copy(), z, z.copy(), z.copy(), np.ones(n, np.float32))
e.bbox(); e.morton(); e.sort()  # caller-order input: runs arrive unsorted
e.sync()
lib = ctypes.CDLL(os.environ["BH_LIB_PATH"])
buf = np.zeros((256, 16), dtype=np.uint64)
assert lib.bh_debug_ls_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
size, npass = buf[:, 4].astype(np.int64), buf[:, 6].astype(np.int64)
ok = size > 0
path = buf[:, 7].astype(np.int64)
names = {0: "full set of passes at once", 1: "window only", 2: "window + neighbour exchanges", 3: "window, exchanges not finished, full set"}
print("paths", {names[int(k)]: int(v) for k, v in zip(*np.unique(path[ok], return_counts=True))})
print("buckets", ok.sum(), "passes histogram", dict(zip(*np.unique(npass[ok], return_counts=True))), "slow", e.stats().sort_slow_buckets)
