"""How much of the X4 all-to-all is padding, and how steady the per-pair needs are (bh_dd_needs_matrix): P ranks on one
GPU through bh_group; per step the sum of the needs against world x world x stride, and the largest growth of any
pair's need since the step before (what a per-pair transfer size taken from the previous step would have to cover).
    python tools/dd_needs.py [--world 8] [--n 8000000] [--steps 40]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhpkg

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--n", type=int, default=8_000_000)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
pkg = bhpkg.load()
from nbody_barnes_hut_cuda_amd import _lib as L
ic = pkg.plummer(a.n, seed=42)
g = C.c_void_p()
dev = (C.c_int * a.world)(*([0] * a.world))
assert L.lib.bh_create_group(C.byref(g), a.world, dev, a.n, None, None, 0) == 0
assert L.lib.bh_group_upload(g, *[np.ascontiguousarray(x).ctypes.data_as(L._F) for x in ic]) == 0
prev = None
info = L.BhRankInfo()
P = a.world
worst_growth, worst_abs = 0.0, 0
for s in range(a.steps):
    stride_used = None
    L.lib.bh_rank_get_info(L.lib.bh_group_rank(g, 0), C.byref(info))
    stride_next = info.stride
    assert L.lib.bh_step_group(g, 1) == 0
    m = np.zeros(P * P, np.int32)
    assert L.lib.bh_dd_needs_matrix(L.lib.bh_rank_ctx(L.lib.bh_group_rank(g, 0)), m.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    m = m.reshape(P, P)
    off = m[~np.eye(P, dtype=bool)]
    L.lib.bh_rank_get_info(L.lib.bh_group_rank(g, 0), C.byref(info))
    recv = info.x4_recv_kb / 1024.0
    line = f"step {s + 1:3d}: X4 received by rank 0 {recv:6.1f} MiB  stride {stride_next:7d}  needs min {off.min():7d} mean {off.mean():9.0f} max {off.max():7d}  sum / (P (P-1) stride) = {off.sum() / (P * (P - 1) * stride_next):.3f}  retries so far {info.let_retries}"
    if prev is not None:
        gr = (m - prev)[~np.eye(P, dtype=bool)]
        rel = gr / np.maximum(prev[~np.eye(P, dtype=bool)], 1)
        line += f"  largest growth of a pair {gr.max():6d} records = {100 * rel.max():5.1f} %"
        if s > 2:
            worst_growth = max(worst_growth, float(rel.max()))
            worst_abs = max(worst_abs, int(gr.max()))
    print(line, flush=True)
    prev = m
print(f"largest growth of any pair after the first steps: {100 * worst_growth:.1f} % / {worst_abs} records")
print("needs matrix of the last step (row = sender, column = receiver):")
print(prev)
L.lib.bh_destroy_group(g)
