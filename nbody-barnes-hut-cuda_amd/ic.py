"""Synthetic initial conditions (host side, bh_ic_* of libbh.so; no GPU needed).

The reference draws a rotating disc with srand(42)/rand() (nbody_v5_bench.cu:294-308);
BASELINE.json asks for Plummer spheres.  Both use the library's counter-based RNG.
"""
import ctypes as C

import numpy as np

from ._lib import lib

_F = C.POINTER(C.c_float)


def _alloc(n):
    return [np.empty(n, dtype=np.float32) for _ in range(7)]


def _ptrs(arrs):
    return [a.ctypes.data_as(_F) for a in arrs]


def plummer(n, seed=42, a=400.0, G=0.5):
    """-> x, y, z, vx, vy, vz, m  (float32 arrays of length n)."""
    arrs = _alloc(n)
    st = lib.bh_ic_plummer(int(n), int(seed), float(a), float(G), *_ptrs(arrs))
    if st != 0:
        raise ValueError(lib.bh_strerror(st).decode())
    return tuple(arrs)


def disc(n, seed=42, G=0.5):
    """The reference's disc by formula (nbody_v5_bench.cu:297-307)."""
    arrs = _alloc(n)
    st = lib.bh_ic_disc(int(n), int(seed), float(G), *_ptrs(arrs))
    if st != 0:
        raise ValueError(lib.bh_strerror(st).decode())
    return tuple(arrs)


def disc_msvc(n, seed=42, G=0.5):
    """The disc exactly as the reference binary draws it: srand(seed) + MSVC rand(), call order of
    nbody_v5_bench.cu:294-308 (for diffing literal_force trajectories against nbody_v5_bench.exe)."""
    arrs = _alloc(n)
    st = lib.bh_ic_disc_msvc(int(n), int(seed), float(G), *_ptrs(arrs))
    if st != 0:
        raise ValueError(lib.bh_strerror(st).decode())
    return tuple(arrs)
