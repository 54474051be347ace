"""design study driver for sim_grouping.c (CPU only): VALU work of the shared walk under different groupings.
usage: sim_groups.py [n] [theta] [sample_stride]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bhpkg, oracle as O
from helpers import oracle_pipeline
so = os.path.join(ROOT, "tools", "bin", "libsim_grouping.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "tests", "studies", "sim_grouping.c"), "-o", so, "-lm"])
L = C.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 16
curve = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pkg = bhpkg.load()
ic = pkg.plummer(n, seed=42)
p = O.params(theta=theta, key_curve=curve)
d = oracle_pipeline(O, ic, p)
rec, xyzm = np.ascontiguousarray(d["rec"]), np.ascontiguousarray(d["xyzm"])
sk = d["sorted_keys"].astype(np.uint64)
# shared leading 3-bit digits of consecutive keys (63-bit keys: 21 digits)
x = sk[1:] ^ sk[:-1]
nb = np.zeros(n - 1, np.int64)
nz = x != 0
nb[nz] = np.floor(np.log2(x[nz].astype(np.float64))).astype(np.int64) + 1   # bit length (approximate above 2^53: fine)
shared = np.concatenate([[0], (63 - nb) // 3]).astype(np.int32)  # shared[j]: boundary between j-1 and j


def run(gs, gc, sub=64):
    gs = np.ascontiguousarray(gs, np.int32); gc = np.ascontiguousarray(gc, np.int32)
    out = np.zeros(8, np.uint64)
    L.sim_groups(C.c_void_p(rec.ctypes.data), C.c_void_p(xyzm.ctypes.data), C.c_void_p(gs.ctypes.data),
                 C.c_void_p(gc.ctypes.data), len(gs), C.c_float(theta), C.c_float(p.eps2), sub, C.c_void_p(out.ctypes.data))
    return out


def fixed(g):
    s = np.arange(0, n, g)
    return s, np.minimum(g, n - s)


def greedy(minfill, maxfill=64):
    gs, gc = [], []
    s = 0
    sh = shared
    while s < n:
        hi = min(s + maxfill, n)
        lo = min(s + minfill, n)
        if hi >= n:
            e = n
        else:
            cand = sh[lo:hi + 1]          # boundaries e = lo..hi
            k = len(cand) - 1 - int(np.argmin(cand[::-1]))   # last minimum
            e = lo + k
        gs.append(s); gc.append(e - s); s = e
    return np.array(gs), np.array(gc)


def report(name, gs, gc, sub=64):
    sel = slice(0, None, stride)
    bodies = gc[sel].sum()
    o = run(gs[sel], gc[sel], sub)
    waves = len(gs[sel])
    B, P, PH, LE, R, PS = [float(v) for v in o[:6]]
    print(f"{name:28s} waves/1M {len(gs)*1e6/n/1e3:6.2f}k fill {gc.mean()/ (sub):5.3f}  per body: pairs {P/bodies:7.2f} "
          f"pairs(sub) {PS/bodies:7.2f} pairs(half/quarter dual) {PH/bodies:7.2f} blocks {B/bodies:6.2f}  "
          f"lane eff {LE/(R*sub):5.3f}  recs/wave {R/waves:6.0f}")


print(f"n={n} theta={theta} curve={curve}")
report("fixed 64", *fixed(64))
report("fixed 32 (2 pairs/instr)", *fixed(32), sub=32)
report("fixed 16 (4 pairs/instr)", *fixed(16), sub=16)
for mf in (56, 48, 40, 32):
    report(f"greedy cut, fill {mf}..64", *greedy(mf))


# ---- regrouping inside Hilbert tiles: balanced kd splits (largest-extent axis, median) down to 64-body groups
def kd_regroup(T):
    perm = np.arange(n)
    pos = xyzm[:, :3]
    out = np.empty(n, np.int64)
    for t0 in range(0, n - n % T, T * stride):     # only the sampled tiles matter
        idx = [np.arange(t0, t0 + T)]
        while len(idx[0]) > 64:
            nxt = []
            for a in idx:
                p = pos[a]
                ax = int(np.argmax(p.max(0) - p.min(0)))
                o = np.argsort(p[:, ax], kind="stable")
                h = len(a) // 2
                nxt += [a[o[:h]], a[o[h:]]]
            idx = nxt
        out[t0:t0 + T] = np.concatenate(idx)
    return out


for T in (128, 512, 4096):
    tiles = np.arange(0, n - n % T, T * stride)
    perm = kd_regroup(T)
    sel = np.concatenate([np.arange(t, t + T) for t in tiles])
    x2 = np.ascontiguousarray(xyzm[perm[sel]])
    gs = np.arange(0, len(sel), 64); gc = np.full(len(gs), 64)
    keep = xyzm
    xyzm = x2
    o = run(gs, gc)
    xyzm = keep
    x1 = np.ascontiguousarray(keep[sel]); xyzm = x1
    o0 = run(gs, gc)
    xyzm = keep
    print(f"kd regroup inside {T}-body Hilbert tiles: pairs/body {float(o[1])/len(sel):7.2f}  (plain Hilbert groups on the same tiles {float(o0[1])/len(sel):7.2f})")
