"""design study: per-block timeline of pairs_kernel / emit_kernel (library built with -DBH_TREE_TRACE,
selected with BH_LIB_PATH)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
e = pkg.Engine(n)
ic = pkg.plummer(n, seed=42)
e.upload(*ic)
if os.environ.get("BH_TRACE_BUILD_ONLY"):  # safe when the build is deliberately broken for an experiment
    for it in range(4):
        e.upload(*ic)
        e.bbox(); e.morton(); e.sort(); e.build()
else:
    e.step(6)
e.sync()
lib = ctypes.CDLL(os.environ["BH_LIB_PATH"])
buf = np.zeros((2, 8192, 12), dtype=np.uint64)
assert lib.bh_debug_tree_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
nt = (n + 1023) // 1024
for kern, name, labels, ns in ((0, "pairs", ["window", "phase 1", "wide cells", "tile scan"], 5),
                               (1, "emit", ["window", "phase 1", "wide cells"], 4)):
    t = buf[kern, :nt, :ns].astype(np.int64)
    nw = buf[kern, :nt, 6].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    d = np.diff(us, axis=1)
    print(f"{name}: blocks {nt}, first start..last start {us[:,0].max():.2f} us, last end {us[:,-1].max():.2f} us; "
          f"wide cells per block median {int(np.median(nw))} max {nw.max()} (blocks with any: {(nw>0).sum()})")
    for k, lab in enumerate(labels):
        print(f"   {lab:11s} median {np.median(d[:,k]):6.2f}  p90 {np.percentile(d[:,k],90):6.2f}  max {d[:,k].max():6.2f}")
    print(f"   block total median {np.median(us[:,-1]-us[:,0]):6.2f}  max {(us[:,-1]-us[:,0]).max():6.2f}")
    if kern == 0:  # wide-cell phase of pairs in detail (blocks with wide cells only)
        sel = nw > 0
        tw = (buf[0, :nt][sel][:, [2, 8, 9, 10, 3]].astype(np.int64)) / 100.0
        dw = np.diff(tw, axis=1)
        for k, lab in enumerate(["stage samples", "k[j], k[j-1]", "lower bounds", "rest"]):
            print(f"      wide: {lab:14s} median {np.median(dw[:,k]):6.2f}  p90 {np.percentile(dw[:,k],90):6.2f}  max {dw[:,k].max():6.2f}")
    if kern == 1:  # phase 1 of emit in detail (thread 0 of every block)
        tw = (buf[1, :nt][:, [1, 8, 9, 10, 2]].astype(np.int64)) / 100.0
        dw = np.diff(tw, axis=1)
        for k, lab in enumerate(["loads pn/cb/pa/pb", "window queries", "parents' offsets", "stores + sync"]):
            print(f"      phase 1: {lab:18s} median {np.median(dw[:,k]):6.2f}  p90 {np.percentile(dw[:,k],90):6.2f}  max {dw[:,k].max():6.2f}")
    order = np.argsort(us[:, 0])
    print("   start-time quantiles (us):", np.round(np.percentile(us[:, 0], [25, 50, 75, 90, 100]), 2))
