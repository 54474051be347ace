"""What the force passes of the domain-decomposed step walk (round 5): P ranks on one GPU through bh_group, a few
steps, then the counted walk (bh_dd_walk_stats) of every rank over its stitched pool — blocks popped and record pairs
evaluated per 64-body group for the one-pass tree, and with --split for the own-pieces and the remote pass.
    python tools/dd_walk_stats.py [--world 8] [--n 8000000] [--steps 3] [--split]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhpkg

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--n", type=int, default=8_000_000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--split", action="store_true")
ap.add_argument("--split-pct", type=int, default=0)
a = ap.parse_args()
pkg = bhpkg.load()
from nbody_barnes_hut_cuda_amd import _lib as L
ic = pkg.plummer(a.n, seed=42)
o = L.BhRankOpts(); L.lib.bh_rank_default_opts(C.byref(o)); o.split = int(a.split); o.split_pct = a.split_pct
g = C.c_void_p()
dev = (C.c_int * a.world)(*([0] * a.world))
assert L.lib.bh_create_group(C.byref(g), a.world, dev, a.n, None, C.byref(o), 0) == 0
assert L.lib.bh_group_upload(g, *[np.ascontiguousarray(x).ctypes.data_as(L._F) for x in ic]) == 0
assert L.lib.bh_step_group(g, a.steps) == 0 and L.lib.bh_group_sync(g) == 0
print(f"{a.world} ranks x {a.n // a.world} bodies, after {a.steps} steps, split={a.split}")
print("rank which      waves  blocks/wave  pairs/wave  masked/wave  lane spills/wave  mean wave cycles  max")
tot = {}
for q in range(a.world):
    ctx = L.lib.bh_rank_ctx(L.lib.bh_group_rank(g, q))
    for which in ((0, 1) if a.split else (0,)):
        ws = L.BhWalkStats()
        st = L.lib.bh_dd_walk_stats(ctx, which, C.byref(ws))
        assert st == 0, st
        name = ("remote pass" if which == 0 else "own pass") if a.split else "one pass"
        w = max(1, ws.waves)
        print(f"{q:4d} {name:11s} {ws.waves:6d} {ws.blocks / w:11.1f} {ws.pairs / w:11.1f} {ws.masked_pairs / w:11.1f} "
              f"{ws.lane_spills / w:13.1f} {ws.wave_cycles_mean:16.0f} {ws.wave_cycles_max:10.0f}")
        t = tot.setdefault(name, [0, 0, 0])
        t[0] += ws.waves; t[1] += ws.blocks; t[2] += ws.pairs
if a.split:
    part = 0 < a.split_pct < 100
    print("rank   own pass alone ms   remote pass alone ms   " + ("one pass of the other bodies   that and the remote pass at once"
                                                                   if part else "both at once ms (two streams)"))
    for q in range(a.world):
        ctx = L.lib.bh_rank_ctx(L.lib.bh_group_rank(g, q))
        ms = (C.c_float * 4)()
        assert L.lib.bh_dd_pass_times(ctx, ms) == 0
        print(f"{q:4d} {ms[0]:14.3f} {ms[1]:20.3f} {ms[2]:18.3f}" + (f" {ms[3]:30.3f}" if part else ""))
for k, t in tot.items():
    print(f"all ranks, {k}: blocks/wave {t[1] / t[0]:.1f} pairs/wave {t[2] / t[0]:.1f}")
L.lib.bh_destroy_group(g)
