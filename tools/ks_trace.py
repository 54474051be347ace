"""design study: per-block timeline of keys_split_kernel (library built with -DBH_OS_TRACE, BH_LIB_PATH)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
e = pkg.Engine(n)
e.upload(*pkg.plummer(n, seed=42))
e.step(6)
e.sync()
lib = ctypes.CDLL(os.environ["BH_LIB_PATH"])
buf = np.zeros((1024, 8), dtype=np.uint64)
assert lib.bh_debug_ks_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
ok = buf[:, 0] > 0
t = buf[ok, :6].astype(np.int64)
us = (t - t[:, 0].min()) / 100.0
d = np.diff(us, axis=1)
print(f"keys_split: blocks {ok.sum()}, start spread {us[:,0].max():.2f} us, last end {us[:,5].max():.2f} us")
for k, lab in enumerate(["tile keys (load, key, store)", "splitter keys + median", "rank sort", "bucket guess + LDS counts",
                         "global bucket atomics"]):
    print(f"   {lab:30s} median {np.median(d[:,k]):6.2f}  p90 {np.percentile(d[:,k],90):6.2f}  max {d[:,k].max():6.2f}")
