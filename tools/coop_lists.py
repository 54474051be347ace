"""Longest level list any wave of the cooperative walk writes (study build: BH_LIB_PATH=tools/bin/libs/study.so,
BH_COOP_SUBSH=13 so that nothing overflows), per theta and body count: what the list capacity classes of
force_coop_subsh are sized by.   python tools/coop_lists.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402

pkg = bhpkg.load()
fn = pkg.lib.bh_debug_devinfo
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p]
for ic_name in ("plummer", "disc"):
    for n in (65536, 1000000):
        ic = pkg.plummer(n, seed=42) if ic_name == "plummer" else pkg.disc(n, seed=42)
        for theta in (0.8, 0.6, 0.5, 0.45, 0.4, 0.35, 0.3, 0.25, 0.2):
            row = []
            for K in (4, 8) if n < 200000 else (4,):
                e = pkg.Engine(n, theta=theta, force_coop=K, force_group=64)
                e.upload(*ic)
                e.tree_stages()
                e.force()
                e.sync()
                out = np.zeros(8, np.int32)
                assert fn(e._h, out.ctypes.data_as(C.c_void_p)) == 0
                row.append(f"K={K}: longest list {out[5 + 1] if False else out[6]} redo {e.stats().force_redo_waves}")
                e.close()
            print(f"{ic_name} n={n} theta={theta}: " + "; ".join(row), flush=True)
