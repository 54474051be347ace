"""ctypes binding of the CPU oracle (oracle/bh_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from the product package.  PARITY UNPINNED (see bh_oracle.h): the
reference has no golden vectors and cannot be built in this image.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libbh_oracle.so")

ORDER_PREORDER, ORDER_BATCHED, ORDER_PREORDER_RECURSIVE = 0, 1, 2
KIND_BODY, KIND_INTERNAL, KIND_MULTI, KIND_PAD = 0, 1, 2, 3

NODE_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("m", "f4"), ("s", "f4"),
                       ("first", "i4"), ("count", "i4"), ("kind", "i4")])


class Params(C.Structure):
    _fields_ = [("G", C.c_float), ("theta", C.c_float), ("dt", C.c_float), ("eps2", C.c_float),
                ("max_speed", C.c_float), ("leaf_cap", C.c_int32), ("max_depth", C.c_int32),
                ("key_bits", C.c_int32), ("compress", C.c_int32), ("key_curve", C.c_int32)]


def build(force=False):
    """Compile the oracle with gcc (needs oracle/Makefile; a no-op when up to date)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "bh_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_I32 = C.POINTER(C.c_int32)
_U64 = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.bho_default_params.argtypes = [C.POINTER(Params)]
        L.bho_bbox.argtypes = [_F, _F, _F, C.c_int, _F]
        L.bho_morton30.argtypes = [_F, _F, _F, _F, C.c_int, _U32, _I32]
        L.bho_keys.argtypes = [_F, _F, _F, _F, C.c_int, C.c_int, C.c_int, _U64]
        L.bho_hilbert_index.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.bho_hilbert_index.restype = C.c_uint64
        L.bho_hilbert_cell.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
        L.bho_sort.argtypes = [_U64, C.c_int, _U64, _I32]
        L.bho_root_edge.argtypes = [_F]
        L.bho_root_edge.restype = C.c_float
        L.bho_build.argtypes = [_U64, C.c_int, C.POINTER(Params), C.c_float, C.c_void_p, _I32, _I32,
                                C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bho_build.restype = C.c_int
        L.bho_com.argtypes = [C.c_void_p, _I32, _I32, C.c_int, _F]
        L.bho_force.argtypes = [C.c_void_p, _F, C.c_int, C.c_int, C.POINTER(Params), C.c_int, _F,
                                _U32, _U32, _U32, C.c_int]
        L.bho_integrate.argtypes = [_F, _F, _F, C.c_int, C.POINTER(Params)]
        L.bho_direct_f64.argtypes = [_F, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.POINTER(C.c_double), C.c_int]
        L.bho_create.argtypes = [C.c_int, C.POINTER(Params)]
        L.bho_create.restype = C.c_void_p
        L.bho_destroy.argtypes = [C.c_void_p]
        L.bho_upload.argtypes = [C.c_void_p] + [_F] * 7
        L.bho_step.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bho_download.argtypes = [C.c_void_p] + [_F] * 6
        L.bho_download_acc.argtypes = [C.c_void_p] + [_F] * 3
        L.bho_last_counts.argtypes = [C.c_void_p, _U64, _U64, _U64, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bho_last_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.bho_group_stats.argtypes = [C.c_void_p, _F, C.c_int, C.POINTER(Params), C.c_int, C.c_int, _U64, _U64,
                                      _U64, _U64]
        L.bho_max_threads.restype = C.c_int
        _lib = L
    return _lib


def params(**kw):
    p = Params()
    lib().bho_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_F)


def max_threads():
    return lib().bho_max_threads()


# ---------------------------------------------------------------- stage functions
def bbox(x, y, z):
    x, y, z = _f(x), _f(y), _f(z)
    b = np.empty(6, np.float32)
    lib().bho_bbox(_fp(x), _fp(y), _fp(z), len(x), _fp(b))
    return b


def morton30(x, y, z, bounds):
    x, y, z, bounds = _f(x), _f(y), _f(z), _f(bounds)
    codes = np.empty(len(x), np.uint32)
    idx = np.empty(len(x), np.int32)
    lib().bho_morton30(_fp(x), _fp(y), _fp(z), _fp(bounds), len(x), codes.ctypes.data_as(_U32),
                       idx.ctypes.data_as(_I32))
    return codes, idx


def keys(x, y, z, bounds, key_bits=63, key_curve=0):
    x, y, z, bounds = _f(x), _f(y), _f(z), _f(bounds)
    k = np.empty(len(x), np.uint64)
    lib().bho_keys(_fp(x), _fp(y), _fp(z), _fp(bounds), len(x), int(key_bits), int(key_curve),
                   k.ctypes.data_as(_U64))
    return k


def hilbert_index(x, y, z):
    """cell (x, y, z) of the 2^21-per-axis grid -> its number along the Hilbert curve"""
    return int(lib().bho_hilbert_index(int(x), int(y), int(z)))


def hilbert_cell(index):
    out = (C.c_uint32 * 3)()
    lib().bho_hilbert_cell(int(index), out)
    return int(out[0]), int(out[1]), int(out[2])


def sort(k):
    k = np.ascontiguousarray(k, dtype=np.uint64)
    sk = np.empty_like(k)
    perm = np.empty(len(k), np.int32)
    lib().bho_sort(k.ctypes.data_as(_U64), len(k), sk.ctypes.data_as(_U64), perm.ctypes.data_as(_I32))
    return sk, perm


def root_edge(bounds):
    return float(lib().bho_root_edge(_fp(_f(bounds))))


def build_tree(sorted_keys, p, s0):
    """-> (rec, er_lo, er_hi, n_internal, max_level); rec is a NODE_DTYPE array (topology only)."""
    sk = np.ascontiguousarray(sorted_keys, dtype=np.uint64)
    n = len(sk)
    cap = 3 * n + 8
    rec = np.zeros(cap, NODE_DTYPE)
    lo = np.zeros(cap, np.int32)
    hi = np.zeros(cap, np.int32)
    ni, ml = C.c_int(0), C.c_int(0)
    E = lib().bho_build(sk.ctypes.data_as(_U64), n, C.byref(p), C.c_float(s0), rec.ctypes.data,
                        lo.ctypes.data_as(_I32), hi.ctypes.data_as(_I32), cap, C.byref(ni), C.byref(ml))
    if E < 0:
        raise RuntimeError("oracle tree pool overflow")
    return rec[:E].copy(), lo[:E].copy(), hi[:E].copy(), ni.value, ml.value


def com(rec, er_lo, er_hi, xyzm):
    rec = np.ascontiguousarray(rec)
    xyzm = _f(xyzm)
    lib().bho_com(rec.ctypes.data, er_lo.ctypes.data_as(_I32), er_hi.ctypes.data_as(_I32), len(rec),
                  _fp(xyzm))
    return rec


def force(rec, xyzm, p, order=ORDER_PREORDER, lo=0, hi=None, counters=True, nthreads=0):
    """-> acc[n,4] (rows outside [lo,hi) are zero), V, O, P."""
    rec = np.ascontiguousarray(rec)
    xyzm = _f(xyzm)
    n = xyzm.shape[0]
    hi = n if hi is None else hi
    acc = np.zeros((n, 4), np.float32)
    V = np.zeros(n, np.uint32)
    O = np.zeros(n, np.uint32)
    P = np.zeros(n, np.uint32)
    lib().bho_force(rec.ctypes.data, _fp(xyzm), int(lo), int(hi), C.byref(p), int(order), _fp(acc),
                    V.ctypes.data_as(_U32) if counters else None,
                    O.ctypes.data_as(_U32) if counters else None,
                    P.ctypes.data_as(_U32) if counters else None, int(nthreads))
    return acc, V, O, P


def group_stats(rec, xyzm, p, group=64, stride=1):
    """Work of the shared traversal of `group` consecutive sorted bodies (the GPU kernel's scheme), summed over
    every stride-th group: -> dict(records, pops, leaf_bodies, groups)."""
    rec = np.ascontiguousarray(rec)
    xyzm = _f(xyzm)
    out = [C.c_uint64() for _ in range(4)]
    lib().bho_group_stats(rec.ctypes.data, _fp(xyzm), xyzm.shape[0], C.byref(p), int(group), int(stride),
                          *[C.byref(o) for o in out])
    return dict(records=out[0].value, pops=out[1].value, leaf_bodies=out[2].value, groups=out[3].value)


def integrate(xyzm, vel3, acc4, p):
    xyzm, vel3, acc4 = _f(xyzm).copy(), _f(vel3).copy(), _f(acc4)
    lib().bho_integrate(_fp(xyzm), _fp(vel3), _fp(acc4), xyzm.shape[0], C.byref(p))
    return xyzm, vel3


def direct_f64(xyzm, G, eps2, lo=0, hi=None, nthreads=0):
    xyzm = _f(xyzm)
    n = xyzm.shape[0]
    hi = n if hi is None else hi
    out = np.zeros((hi - lo, 3), np.float64)
    lib().bho_direct_f64(_fp(xyzm), n, int(lo), int(hi), C.c_float(G), C.c_float(eps2),
                         out.ctypes.data_as(C.POINTER(C.c_double)), int(nthreads))
    return out


# ---------------------------------------------------------------- whole-step state
class Oracle:
    """CPU mirror of the engine's step loop (simulationStep, nbody_v5_bench.cu:255-283)."""

    def __init__(self, n, p=None, **kw):
        self.n = int(n)
        self.p = p if p is not None else params(**kw)
        self._h = lib().bho_create(self.n, C.byref(self.p))

    def close(self):
        if self._h:
            lib().bho_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, x, y, z, vx, vy, vz, m):
        arrs = [_f(a) for a in (x, y, z, vx, vy, vz, m)]
        lib().bho_upload(self._h, *[_fp(a) for a in arrs])

    def step(self, steps=1, order=ORDER_PREORDER, nthreads=0):
        for _ in range(int(steps)):
            lib().bho_step(self._h, int(order), int(nthreads))

    def download(self):
        out = [np.empty(self.n, np.float32) for _ in range(6)]
        lib().bho_download(self._h, *[_fp(a) for a in out])
        return tuple(out)

    def download_acc(self):
        out = [np.empty(self.n, np.float32) for _ in range(3)]
        lib().bho_download_acc(self._h, *[_fp(a) for a in out])
        return tuple(out)

    def counts(self):
        V, O, P = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ni, ne, ml = C.c_int(), C.c_int(), C.c_int()
        lib().bho_last_counts(self._h, C.byref(V), C.byref(O), C.byref(P), C.byref(ni), C.byref(ne),
                              C.byref(ml))
        return dict(V=V.value, O=O.value, P=P.value, n_internal=ni.value, n_entries=ne.value,
                    max_level=ml.value)

    def times(self):
        t = (C.c_double * 7)()
        lib().bho_last_times(self._h, t)
        return dict(zip(["bbox", "keys", "sort", "build", "com", "force", "integrate"], list(t)))
