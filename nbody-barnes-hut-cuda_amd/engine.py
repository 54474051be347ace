"""Engine: one Barnes-Hut context on one MI355X (wraps bh_ctx of include/bh.h)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, BhParams, BhNode, BhStats

KIND_BODY, KIND_INTERNAL, KIND_MULTI, KIND_PAD = 0, 1, 2, 3
_F = C.POINTER(C.c_float)


class BhError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        super().__init__(f"{where}: {lib.bh_strerror(status).decode()} ({status})")


def default_params(**kw):
    """bh_params with the reference's constants (nbody_v5_bench.cu:14-18), overridable."""
    p = BhParams()
    lib.bh_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(f"bh_params has no field {k}")
        setattr(p, k, v)
    return p


def write_text(path, steps, theta, dt, x, y, z, vx, vy, vz):
    """host-only: -> bh_status"""
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, z, vx, vy, vz)]
    return lib.bh_write_text(path.encode(), len(arrs[0]), int(steps), float(theta), float(dt),
                             *[a.ctypes.data_as(_F) for a in arrs])


def read_text(path):
    """host-only: -> (steps, x, y, z, vx, vy, vz)"""
    n, steps = C.c_int(0), C.c_int(0)
    st = lib.bh_read_text(path.encode(), 0, C.byref(n), C.byref(steps), *([None] * 6))
    if st not in (0, -7):
        raise BhError(st, "bh_read_text")
    arrs = [np.empty(n.value, dtype=np.float32) for _ in range(6)]
    st = lib.bh_read_text(path.encode(), n.value, C.byref(n), C.byref(steps),
                          *[a.ctypes.data_as(_F) for a in arrs])
    if st != 0:
        raise BhError(st, "bh_read_text")
    return (steps.value,) + tuple(arrs)


def read_snapshot(path):
    """host-only: -> (n, steps, BhParams, (x, y, z, vx, vy, vz, m))"""
    n, steps = C.c_int(0), C.c_int(0)
    p = BhParams()
    st = lib.bh_read_snapshot(path.encode(), 0, C.byref(n), C.byref(steps), C.byref(p), *([None] * 7))
    if st != 0:
        raise BhError(st, "bh_read_snapshot")
    arrs = [np.empty(n.value, dtype=np.float32) for _ in range(7)]
    st = lib.bh_read_snapshot(path.encode(), n.value, C.byref(n), C.byref(steps), C.byref(p),
                              *[a.ctypes.data_as(_F) for a in arrs])
    if st != 0:
        raise BhError(st, "bh_read_snapshot")
    return n.value, steps.value, p, tuple(arrs)


def _f32(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != (n,):
        raise ValueError(f"expected float32[{n}], got {a.shape}")
    return a


class Engine:
    """Replaces the reference's file-scope globals + simulationStep() (:31-40, :255-283)."""

    def __init__(self, n, params=None, device=0, stream=None, **kw):
        self.n = int(n)
        self.params = params if params is not None else default_params(**kw)
        self._h = C.c_void_p()
        if stream is None:
            st = lib.bh_create(C.byref(self._h), self.n, C.byref(self.params), int(device))
        else:
            st = lib.bh_create_on_stream(C.byref(self._h), self.n, C.byref(self.params),
                                         int(device), C.c_void_p(int(stream)))
        if st != 0:
            self._h = C.c_void_p()
            raise BhError(st, "bh_create")

    @classmethod
    def adopt(cls, handle, n, params=None):
        """an Engine view of a context somebody else owns (the context of a bh_rank): close() does not destroy it"""
        self = cls.__new__(cls)
        self.n = int(n)
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p(handle)
        self._borrowed = True
        return self

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not getattr(self, "_borrowed", False):
                lib.bh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st, where):
        if st != 0:
            raise BhError(st, where)

    # -- data in (ref:329-335)
    def upload(self, x, y, z, vx, vy, vz, m):
        arrs = [_f32(a, self.n) for a in (x, y, z, vx, vy, vz, m)]
        self._ck(lib.bh_upload(self._h, *[a.ctypes.data_as(_F) for a in arrs]), "bh_upload")

    # -- the step and its stages, reference order (ref:259-282)
    def step(self, steps=1):
        for _ in range(int(steps)):
            self._ck(lib.bh_step(self._h), "bh_step")

    simulationStep = step  # the reference's name (ref:255)

    def bbox(self):
        self._ck(lib.bh_bbox(self._h), "bh_bbox")

    def morton(self):
        self._ck(lib.bh_morton(self._h), "bh_morton")

    def sort(self):
        self._ck(lib.bh_sort(self._h), "bh_sort")

    def build(self):
        self._ck(lib.bh_build(self._h), "bh_build")

    def com(self):
        self._ck(lib.bh_com(self._h), "bh_com")

    def force(self, lo=None, hi=None):
        if lo is None:
            self._ck(lib.bh_force(self._h), "bh_force")
        else:
            self._ck(lib.bh_force_range(self._h, int(lo), int(hi)), "bh_force_range")

    def force_count(self):
        self._ck(lib.bh_force_count(self._h), "bh_force_count")

    def force_walk_stats(self):
        """-> BhWalkStats: what one launch of the default force walk issued (measurement only)."""
        from ._lib import BhWalkStats
        st = BhWalkStats()
        self._ck(lib.bh_force_walk_stats(self._h, C.byref(st)), "bh_force_walk_stats")
        return st

    def force_launch_trace(self):
        """-> uint32[rows, 4]: per wave of the force launch bh_step would make: start, end of its walk (100 MHz clock),
        HW_ID, XCC_ID (measurement only; empty for contexts whose walk has no traced instance)"""
        cap = max(1024, (self.n + 63) // 64 * 8 + 64)
        rows = np.zeros((cap, 4), np.uint32)
        n = C.c_int(0)
        self._ck(lib.bh_force_launch_trace(self._h, rows.ctypes.data_as(C.POINTER(C.c_uint32)), cap, C.byref(n)),
                 "bh_force_launch_trace")
        return rows[:n.value].copy()

    def integrate(self):
        self._ck(lib.bh_integrate(self._h), "bh_integrate")

    def tree_stages(self):
        """bbox -> morton -> sort -> build -> com (everything before the force stage)."""
        self.bbox(); self.morton(); self.sort(); self.build(); self.com()

    def sync(self):
        self._ck(lib.bh_sync(self._h), "bh_sync")

    def set_timing(self, on=True):
        """True / 1: events after every stage of a step; 2: only around the force launch (cheap); 3: that pair on
        every 4th step; False: off"""
        self._ck(lib.bh_set_timing(self._h, on if on in (2, 3) else (1 if on else 0)), "bh_set_timing")

    # -- data out
    def download(self):
        """-> x, y, z, vx, vy, vz in caller (upload) order."""
        out = [np.empty(self.n, dtype=np.float32) for _ in range(6)]
        self._ck(lib.bh_download(self._h, *[a.ctypes.data_as(_F) for a in out]), "bh_download")
        return tuple(out)

    def download_acc(self):
        out = [np.empty(self.n, dtype=np.float32) for _ in range(3)]
        self._ck(lib.bh_download_acc(self._h, *[a.ctypes.data_as(_F) for a in out]), "bh_download_acc")
        return tuple(out)

    def download_bounds(self):
        b = np.empty(6, dtype=np.float32)
        self._ck(lib.bh_download_bounds(self._h, b.ctypes.data_as(_F)), "bh_download_bounds")
        return b

    def download_keys(self):
        k = np.empty(self.n, dtype=np.uint64)
        self._ck(lib.bh_download_keys(self._h, k.ctypes.data_as(C.POINTER(C.c_uint64))), "bh_download_keys")
        return k

    def download_order(self):
        ids = np.empty(self.n, dtype=np.int32)
        self._ck(lib.bh_download_order(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32))), "bh_download_order")
        return ids

    def download_sorted_bodies(self):
        a = np.empty((self.n, 4), dtype=np.float32)
        self._ck(lib.bh_download_sorted_bodies(self._h, a.ctypes.data_as(_F)), "bh_download_sorted_bodies")
        return a

    def download_tree(self):
        """-> structured numpy array of bh_node records (entry 0 = root)."""
        cnt = C.c_int(0)
        self._ck(lib.bh_download_tree(self._h, None, 0, C.byref(cnt)), "bh_download_tree")
        dt = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("m", "f4"), ("s", "f4"),
                       ("first", "i4"), ("count", "i4"), ("kind", "i4")])
        rec = np.empty(cnt.value, dtype=dt)
        self._ck(lib.bh_download_tree(self._h, rec.ctypes.data_as(C.POINTER(BhNode)), cnt.value,
                                      C.byref(cnt)), "bh_download_tree")
        return rec

    def download_counters(self):
        out = [np.empty(self.n, dtype=np.uint32) for _ in range(3)]
        self._ck(lib.bh_download_counters(self._h, *[a.ctypes.data_as(C.POINTER(C.c_uint32)) for a in out]),
                 "bh_download_counters")
        return tuple(out)

    def stats(self):
        s = BhStats()
        self._ck(lib.bh_get_stats(self._h, C.byref(s)), "bh_get_stats")
        return s

    def download_mass(self):
        m = np.empty(self.n, dtype=np.float32)
        self._ck(lib.bh_download_mass(self._h, m.ctypes.data_as(_F)), "bh_download_mass")
        return m

    def export_visual(self):
        """-> (pos[n,3], rgb[n,3]) as the reference viewer's updateVisualsKernel fills its VBOs
        (nbody_v5.cu:278-292), in caller order."""
        pos = np.empty((self.n, 3), dtype=np.float32)
        col = np.empty((self.n, 3), dtype=np.float32)
        self._ck(lib.bh_export_visual(self._h, pos.ctypes.data_as(_F), col.ctypes.data_as(_F)),
                 "bh_export_visual")
        return pos, col

    # -- state dump / restart
    def dump_text(self, path):
        """final-state text dump in the older generation's format (output_bh.txt:1-4)"""
        arrs = self.download()
        self._ck(write_text(path, self.stats().steps, self.params.theta, self.params.dt, *arrs), "bh_write_text")

    def save_snapshot(self, path):
        arrs = list(self.download()) + [self.download_mass()]
        st = lib.bh_write_snapshot(path.encode(), self.n, self.stats().steps, C.byref(self.params),
                                   *[a.ctypes.data_as(_F) for a in arrs])
        self._ck(st, "bh_write_snapshot")

    @classmethod
    def restore(cls, path, device=0, stream=None):
        """new Engine from a binary snapshot (same parameters, bodies uploaded)"""
        n, steps, params, arrs = read_snapshot(path)
        e = cls(n, params=params, device=device, stream=stream)
        e.upload(*arrs)
        return e

    def timing_history(self):
        """-> (ms_force[], ms_step[]) of the most recent timed steps (hipEvent, engine stream)."""
        cap = 256
        f = np.empty(cap, np.float32)
        t = np.empty(cap, np.float32)
        cnt = C.c_int(0)
        self._ck(lib.bh_timing_history(self._h, f.ctypes.data_as(_F), t.ctypes.data_as(_F), cap,
                                       C.byref(cnt)), "bh_timing_history")
        return f[:cnt.value].copy(), t[:cnt.value].copy()

    # -- multi-rank plumbing
    def bind_acc(self, device_ptr):
        """use a caller-owned device buffer (>= n float4) for the accelerations"""
        self._ck(lib.bh_bind_acc(self._h, C.c_void_p(int(device_ptr) if device_ptr else None)), "bh_bind_acc")

    def device_acc(self):
        """(device pointer, bytes) of the float4[n] acceleration buffer (Morton order)."""
        p = C.c_void_p()
        nb = C.c_int64()
        self._ck(lib.bh_device_acc(self._h, C.byref(p), C.byref(nb)), "bh_device_acc")
        return p.value, nb.value

    # ---- domain-decomposed multi-GPU stepping (include/bh.h bh_dd_*; driven by dist.DomainStepper) ----
    @staticmethod
    def dd_query(n_cap, world, mig_cap, let_cap):
        s = _lib.BhDdSizes()
        st = lib.bh_dd_query(int(n_cap), int(world), int(mig_cap), int(let_cap), C.byref(s))
        if st != 0:
            raise BhError(st, "bh_dd_query")
        return s

    def dd_init(self, world, rank, n_total, mig_cap, let_cap, pool_ptr, pool_records):
        self._ck(lib.bh_dd_init(self._h, int(world), int(rank), int(n_total), int(mig_cap), int(let_cap),
                                C.c_void_p(int(pool_ptr)), int(pool_records)), "bh_dd_init")

    def dd_upload(self, x, y, z, vx, vy, vz, m, ids):
        n_loc = len(x)
        arrs = [_f32(a, n_loc) for a in (x, y, z, vx, vy, vz, m)]
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self._ck(lib.bh_dd_upload(self._h, n_loc, *[a.ctypes.data_as(_lib._F) for a in arrs],
                                  ids.ctypes.data_as(C.POINTER(C.c_int32))), "bh_dd_upload")
        self.n = n_loc

    def dd_cube_pack(self, send_ptr):
        self._ck(lib.bh_dd_cube_pack(self._h, C.c_void_p(int(send_ptr))), "bh_dd_cube_pack")

    def dd_cube_apply(self, gathered_ptr):
        self._ck(lib.bh_dd_cube_apply(self._h, C.c_void_p(int(gathered_ptr))), "bh_dd_cube_apply")

    def dd_migrate_pack(self, send_ptr, limit):
        self._ck(lib.bh_dd_migrate_pack(self._h, C.c_void_p(int(send_ptr)), int(limit)), "bh_dd_migrate_pack")

    def dd_migrate_apply(self, gathered_ptr, limit):
        """(bodies now held, another round needed, most emigrants found on any rank)"""
        n_loc, more, most = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(lib.bh_dd_migrate_apply(self._h, C.c_void_p(int(gathered_ptr)), int(limit), C.byref(n_loc),
                                         C.byref(more), C.byref(most)), "bh_dd_migrate_apply")
        self.n = n_loc.value
        return self.n, bool(more.value), most.value

    def dd_tree(self, send_ptr):
        self._ck(lib.bh_dd_tree(self._h, C.c_void_p(int(send_ptr))), "bh_dd_tree")

    def dd_let_pack(self, gathered_x3_ptr, send_ptr, stride):
        self._ck(lib.bh_dd_let_pack(self._h, C.c_void_p(int(gathered_x3_ptr)), C.c_void_p(int(send_ptr)),
                                    int(stride)), "bh_dd_let_pack")

    def dd_force_local(self, gathered_x3_ptr):
        self._ck(lib.bh_dd_force_local(self._h, C.c_void_p(int(gathered_x3_ptr))), "bh_dd_force_local")

    def dd_top(self, gathered_x3_ptr, stride):
        self._ck(lib.bh_dd_top(self._h, C.c_void_p(int(gathered_x3_ptr)), int(stride)), "bh_dd_top")

    def dd_force(self):
        self._ck(lib.bh_dd_force(self._h), "bh_dd_force")

    def dd_let_check(self, stride, world):
        """(fits, counts): records every rank needed in the last LET exchange."""
        counts = np.zeros(world, np.int32)
        st = lib.bh_dd_let_check(self._h, int(stride), counts.ctypes.data_as(C.POINTER(C.c_int32)))
        if st not in (0, -7):
            raise BhError(st, "bh_dd_let_check")
        return st == 0, counts

    def dd_set_let_mode(self, mode):
        """0: X4 = all-gather of one union segment; 1: per-destination segments exchanged with an all-to-all"""
        self._ck(lib.bh_dd_set_let_mode(self._h, int(mode)), "bh_dd_set_let_mode")

    # one call per phase group (bh_dd_phase_*): what DomainStepper.step uses
    def dd_phase_migrate(self, gathered_x1_ptr, send_x2_ptr, limit):
        self._ck(lib.bh_dd_phase_migrate(self._h, C.c_void_p(int(gathered_x1_ptr)), C.c_void_p(int(send_x2_ptr)),
                                         int(limit)), "bh_dd_phase_migrate")

    def dd_phase_tree(self, gathered_x2_ptr, limit, send_x3_ptr):
        """(bodies now held, another migration round needed, most emigrants found on any rank)"""
        n_loc, more, most = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(lib.bh_dd_phase_tree(self._h, C.c_void_p(int(gathered_x2_ptr)), int(limit),
                                      C.c_void_p(int(send_x3_ptr)), C.byref(n_loc), C.byref(more), C.byref(most)),
                 "bh_dd_phase_tree")
        self.n = n_loc.value
        return self.n, bool(more.value), most.value

    def dd_phase_let(self, gathered_x3_ptr, send_x4_ptr, stride, own_pass):
        self._ck(lib.bh_dd_phase_let(self._h, C.c_void_p(int(gathered_x3_ptr)), C.c_void_p(int(send_x4_ptr)),
                                     int(stride), 1 if own_pass else 0), "bh_dd_phase_let")

    def dd_phase_force(self, gathered_x3_ptr, stride, world):
        """(fits, counts): top tree + remote (or whole) force pass + the records every rank needed in X4"""
        counts = np.zeros(world, np.int32)
        fits = C.c_int(0)
        self._ck(lib.bh_dd_phase_force(self._h, C.c_void_p(int(gathered_x3_ptr)), int(stride),
                                       counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(fits)), "bh_dd_phase_force")
        return bool(fits.value), counts

    def dd_phase_end(self, send_x1_ptr):
        self._ck(lib.bh_dd_phase_end(self._h, C.c_void_p(int(send_x1_ptr))), "bh_dd_phase_end")

    def dd_info(self):
        """(bodies held, emigrants found in the last step, steps in which the domain boundaries moved, what the last
        step did with them: 0 kept / 1 exact quantiles / 2 sample quantiles) — synchronises; logs and tests"""
        out = np.zeros(8, np.int32)
        self._ck(lib.bh_dd_get_info(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))), "bh_dd_get_info")
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def dd_download(self):
        """local bodies in local Morton order: posm [n,4], vel [n,3], ids [n], acc [n,3]"""
        n = self.n
        posm = np.empty((n, 4), np.float32)
        velid = np.empty((n, 4), np.float32)
        acc = np.empty((n, 4), np.float32)
        self._ck(lib.bh_dd_download(self._h, posm.ctypes.data_as(_lib._F), velid.ctypes.data_as(_lib._F),
                                    acc.ctypes.data_as(_lib._F)), "bh_dd_download")
        ids = velid[:, 3].copy().view(np.int32)
        return posm, velid[:, :3].copy(), ids, acc[:, :3].copy()
