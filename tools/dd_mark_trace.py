"""Where dd_mark_kernel's blocks spend their time (design study; build: tools/mkvariant.sh ddtrace -DBH_DD_TRACE, run with
BH_LIB_PATH=tools/bin/libs/ddtrace.so): 8 ranks x 1M on one GPU through bh_group, a few steps, then the per-block
stamps of the LAST launch (100 MHz wall clock): start, end, and the sums of the three phases of its chunks — a: load +
rank-box test of every record, b: piece-box tests of the candidates (eight lanes per candidate), c: compaction —,
candidates seen."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhpkg

pkg = bhpkg.load()
from nbody_barnes_hut_cuda_amd import _lib as L
world, n = 8, 8_000_000
ic = pkg.plummer(n, seed=42)
g = C.c_void_p()
dev = (C.c_int * world)(*([0] * world))
assert L.lib.bh_create_group(C.byref(g), world, dev, n, None, None, 0) == 0
assert L.lib.bh_group_upload(g, *[np.ascontiguousarray(x).ctypes.data_as(L._F) for x in ic]) == 0
assert L.lib.bh_step_group(g, 4) == 0 and L.lib.bh_group_sync(g) == 0
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3   # the launch looked at: rank R's, replayed alone (bh_rank_replay_force_phase)
ms = C.c_float()
assert L.lib.bh_rank_replay_force_phase(L.lib.bh_group_rank(g, R), 0, 0, 0, 1, C.byref(ms)) == 0
print(f"rank {R}: replayed force phase {ms.value:.3f} ms")
raw = np.zeros((2048, 8), np.uint64)
L.lib.bh_debug_dd_trace.argtypes = [C.c_void_p]
assert L.lib.bh_debug_dd_trace(raw.ctypes.data_as(C.c_void_p)) == 0
t = raw[raw[:, 5] > 0].astype(np.float64)
t0 = t[:, 0].min()
us = lambda v: v / 100.0
print(f"blocks {len(t)}; launch span {us(t[:, 5].max() - t0):.1f} us (last-block scan ends {us(t[:, 6].max() - t0):.1f})")
print(f"block start us: min {us(t[:, 0].min() - t0):.1f} max {us(t[:, 0].max() - t0):.1f}; end: p10 {us(np.percentile(t[:, 5], 10) - t0):.1f} "
      f"median {us(np.median(t[:, 5]) - t0):.1f} max {us(t[:, 5].max() - t0):.1f}")
for k, name in ((1, "a load + rank boxes"), (2, "b candidates x piece boxes"), (3, "c compaction")):
    print(f"  phase {name:28s} per block us: mean {us(t[:, k].mean()):6.1f} p90 {us(np.percentile(t[:, k], 90)):6.1f} max {us(t[:, k].max()):6.1f}")
print(f"  candidates per block: mean {t[:, 4].mean():.0f} max {t[:, 4].max():.0f}")
L.lib.bh_destroy_group(g)
