"""design study: per-bucket timeline of local_sort_kernel (library built with -DBH_OS_TRACE, BH_LIB_PATH)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
e = pkg.Engine(n)
e.upload(*(pkg.disc(n, seed=42) if os.environ.get("BH_TRACE_IC") == "disc" else pkg.plummer(n, seed=42)))
e.step(6)
e.sync()
lib = ctypes.CDLL(os.environ["BH_LIB_PATH"])
buf = np.zeros((256, 16), dtype=np.uint64)
assert lib.bh_debug_ls_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[:, :4].astype(np.int64)
size = buf[:, 4].astype(np.int64)
npass = buf[:, 6].astype(np.int64)  # passes run (4 over the top window; + the full set after a tie)
ok = size > 0
t0 = t[ok, 0].min()
us = (t[ok] - t0) / 100.0
print("buckets", ok.sum(), "size min/median/max", size[ok].min(), int(np.median(size[ok])), size[ok].max())
print("passes min/median/max", npass[ok].min(), int(np.median(npass[ok])), npass[ok].max())
print("start spread %.2f  end max %.2f us" % (us[:, 0].max(), us[:, 3].max()))
d = np.diff(us, axis=1)
for k, name in enumerate(["load", "passes", "write+gather"]):
    print(f"  {name:13s} median {np.median(d[:,k]):6.2f} max {d[:,k].max():6.2f}")
print("  per pass median %.2f us" % np.median(d[:, 1] / np.maximum(npass[ok], 1)))
ph = buf[ok, 8:14].astype(np.int64)
good = ph[:, 0] > 0
dd = np.diff(ph[good], axis=1) / 100.0
for k, name in enumerate(["zero+sync", "rank+sync", "prefix+scan", "scatter+sync", "read back"]):
    print(f"  pass 1 {name:13s} median {np.median(dd[:,k]):6.2f} max {dd[:,k].max():6.2f}")
