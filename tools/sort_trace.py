"""design study: per-tile phase timeline of the onesweep pass kernel (library built with -DBH_OS_TRACE,
selected with BH_LIB_PATH).  Stamps: 0 ticket, 1 counted + ranked, 2 own group summed (digit 0),
3 offsets + staging done, 4 earlier groups summed, 5 end.  Prints, per pass, the spread of start times and the median / max of every phase."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
e = pkg.Engine(n)
ic = pkg.plummer(n, seed=42)
e.upload(*ic)
for it in range(10):
    e.bbox(); e.morton(); e.sort()
e.sync()
lib = ctypes.CDLL(os.environ["BH_LIB_PATH"])
buf = np.zeros((8, 4096, 8), dtype=np.uint64)
rc = lib.bh_debug_os_trace(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
tile = int(os.environ.get("BH_TILE", "4096"))
nt = (n + tile - 1) // tile
for p in range(8):
    t = buf[p, :nt, :6].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    d = np.diff(us, axis=1)
    print(f"pass {p}: tiles {nt} start spread {us[:,0].max():.2f} us, end max {us[:,5].max():.2f} us")
    for k, name in enumerate(["load+rank", "own group", "offs+stage", "earlier grps", "global write"]):
        print(f"   {name:13s} median {np.median(d[:,k]):6.2f}  max {d[:,k].max():6.2f}  (abs end median {np.median(us[:,k+1]):6.2f})")
    if p == 3:
        idx = np.argsort(us[:, 5])[-5:]
        for i in idx: print("   slow tile", i, np.round(us[i], 2))
