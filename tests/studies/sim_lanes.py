"""design study driver for tools/sim_lanes.c (CPU only): lane occupancy of the shared traversal."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
# lives under tests/ because it feeds the ORACLE's tree to the simulation (oracle/ is test infrastructure only)
import bhpkg, oracle as O
from helpers import oracle_pipeline
so = os.path.join(ROOT, "tools", "bin", "libsim_lanes.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "tests", "studies", "sim_lanes.c"), "-o", so, "-lm"])
L = C.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pkg = bhpkg.load()
ic = pkg.plummer(n, seed=42)
p = O.params(theta=theta)
d = oracle_pipeline(O, ic, p)
rec, xyzm = np.ascontiguousarray(d["rec"]), np.ascontiguousarray(d["xyzm"])
thr = np.array([1, 2, 4, 8, 12, 16, 24, 32], np.int32)
hist = np.zeros(65, np.uint64); dense = np.zeros(len(thr), np.uint64); pairs = np.zeros(len(thr), np.uint64)
defers = np.zeros(len(thr), np.uint64)
G = C.c_uint64(); PU = C.c_uint64()
L.sim_lanes(C.c_void_p(rec.ctypes.data), C.c_void_p(xyzm.ctypes.data), n, C.c_float(theta), C.c_float(p.eps2), stride,
            C.c_void_p(hist.ctypes.data), C.c_void_p(thr.ctypes.data), len(thr), C.c_void_p(dense.ctypes.data),
            C.c_void_p(pairs.ctypes.data), C.c_void_p(defers.ctypes.data), C.byref(G), C.byref(PU))
g = G.value
tot = hist.sum()
need = (hist * np.arange(65, dtype=np.uint64)).sum()
print(f"n={n} theta={theta}: groups sampled {g}, records/wave {tot/g:.0f}, lane evals/body {need/g/64:.0f}, "
      f"lane efficiency {need/tot/64:.3f}, pushes/wave {PU.value/g:.0f}")
cum = 0
print("active lanes : share of record evaluations (cumulative)")
for lo, hi in [(1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 63), (64, 64)]:
    s = hist[lo:hi + 1].sum() / tot; cum += s
    print(f"  {lo:2d}-{hi:2d}: {s:6.3f} ({cum:6.3f})")
print("defer threshold T: dense records/wave, deferred pairs/wave, (pairs/64), defers/wave, total at cost ratio 1.0/1.5/2.0 per packed pair-instruction")
for i, T in enumerate(thr):
    dn, pr = dense[i] / g, pairs[i] / g
    print(f"  T={T:2d}: dense {dn:7.0f}  pairs {pr:8.0f} ({pr/64:6.0f})  defers {defers[i]/g:6.0f}   "
          f"total {dn+pr/64:7.0f} / {dn+1.5*pr/64:7.0f} / {dn+2*pr/64:7.0f}")
