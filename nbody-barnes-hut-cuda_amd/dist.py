"""Multi-GPU stepping from Python: one process per GPU (the reference is single-GPU).

DomainStepper — the product's scheme — is a thin binding of `bh_rank` (include/bh.h, csrc/bh_group.hip): every rank
owns the bodies of one interval of the key curve (the splitter keys persist from step to step), builds only their
octree, and per step exchanges  X1 cube + boundary proposals, X2 emigrants, X3 piece descriptors (three all-gathers)
and X4 per-destination locally-essential records (one all-to-all); one force pass over the stitched tree.  The whole
per-step protocol lives in the library (bh_rank_step); this module only chooses how the bytes travel:

  RcclComm    the library's own RCCL transport on the rank's stream (ncclCommInitRank; the unique id is broadcast
              through torch.distributed) — what bench.py uses under torchrun;
  TorchComm   callbacks into torch.distributed (gloo rehearsals on one GPU, CPU protocol tests);
  LocalComm   P ranks as threads of one process sharing a device (bh_hub: device copies) — tests, tools.

ShardedStepper is the round-1 scheme kept for A/B and as the collective fall-back (BH_DIST_MODE=replicated): every
rank holds all bodies and builds the same tree, the force stage is sharded by Morton slab [r*slab, (r+1)*slab), ONE
all-gather of accelerations per step; bit-identical to one rank.
"""
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist


SLAB_ALIGN = 256  # one force-kernel block: slab starts stay wave-aligned, so every wave holds the
                  # same 64 bodies as in the 1-rank run (bit-identical results for every kernel variant)


def slab_size(n, world):
    slab = (n + world - 1) // world
    return (slab + SLAB_ALIGN - 1) // SLAB_ALIGN * SLAB_ALIGN


def slab_bounds(n, world, rank):
    """Morton slab of `rank`: equal 256-aligned slabs, trailing ones may be short or empty."""
    slab = slab_size(n, world)
    lo = min(n, rank * slab)
    hi = min(n, lo + slab)
    return slab, lo, hi


def _into_tensor_ok(group=None):
    """all_gather_into_tensor exists on the "nccl" (= RCCL) backend; gloo (CPU rehearsal) takes the list form.
    Decided from the backend NAME, once — never by catching the failure of a collective: a rank that falls back
    alone after a failed collective would leave the others inside a different one."""
    return str(dist.get_backend(group)).lower() == "nccl"


def all_gather_rows(out, send, group=None, into_tensor=None):
    """out[world*slab, 4] <- concatenation of every rank's send[slab, 4]."""
    if into_tensor is None:
        into_tensor = _into_tensor_ok(group)
    if into_tensor:
        dist.all_gather_into_tensor(out, send, group=group)
    else:
        world = dist.get_world_size(group)
        chunks = list(out.view(world, -1, out.shape[-1]).unbind(0))
        dist.all_gather(chunks, send, group=group)


class ShardedStepper:
    """Drives any engine-like object (tree_stages / force(lo,hi) / integrate, accelerations
    living in `acc`, a [world*slab, 4] float32 torch tensor the engine writes its slab into)."""

    def __init__(self, engine, acc, n, group=None):
        self.e = engine
        self.acc = acc
        self.n = int(n)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.slab, self.lo, self.hi = slab_bounds(self.n, self.world, self.rank)
        self.into_tensor = _into_tensor_ok(group) if dist.is_initialized() else False
        assert acc.shape == (self.world * self.slab, 4) and acc.dtype == torch.float32
        self.send = torch.zeros((self.slab, 4), dtype=acc.dtype, device=acc.device)

    def step(self, steps=1):
        for _ in range(int(steps)):
            self.e.tree_stages()                 # replicated, deterministic
            self.e.force(self.lo, self.hi)       # sharded
            if self.world > 1:                   # the one exchange step
                row0 = self.rank * self.slab
                self.send.copy_(self.acc[row0:row0 + self.slab])
                all_gather_rows(self.acc, self.send, self.group, self.into_tensor)
            self.e.integrate()                   # replicated


def make_gpu_stepper(pkg, n, params=None, device=None, group=None, **kw):
    """Engine on a dedicated torch stream (so RCCL collectives and kernels are ordered by that
    stream) with its acceleration buffer bound to a torch tensor sized for the all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if device is None:
        device = torch.cuda.current_device()
    # a non-default stream: the default stream's handle is 0, which bh_create_on_stream reads as
    # "make your own", and kernels on that private stream would race the collectives
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    slab = slab_size(n, world)
    acc = torch.zeros((world * slab, 4), dtype=torch.float32, device=f"cuda:{device}")
    torch.cuda.synchronize(device)
    eng = pkg.Engine(n, params=params, device=device, stream=stream.cuda_stream, **kw)
    eng.bind_acc(acc.data_ptr())
    eng._acc_keepalive = (acc, stream)
    return eng, ShardedStepper(eng, acc, n, group)


def _round_up(v, a):
    return (int(v) + a - 1) // a * a


# ----------------------------------------------------------------------------------------------
# Domain-decomposed stepping: bh_rank of include/bh.h (csrc/bh_group.hip) + a transport
# ----------------------------------------------------------------------------------------------
def _lib():
    import sys
    return sys.modules[__name__.rsplit(".", 1)[0] + "._lib"]


class _Comm:
    """a filled struct bh_comm + whatever must outlive it"""
    world = rank = 0

    def bh_comm(self):
        raise NotImplementedError

    def attach(self, stepper):
        pass


class TensorComm(_Comm):
    """caller-callback transport (the `callbacks` flavour of struct bh_comm): a subclass moves torch uint8 tensors in
    all_gather(out, send) / all_to_all(out, send).  The rank's buffers are then torch tensors the stepper allocates
    and registers here; a callback finds the tensor a pointer lies in."""

    def __init__(self, world, rank):
        L = _lib()
        self.world, self.rank = world, rank
        self.tensors = []
        self.stream = None
        self.error = None
        self._ag = L.COMM_FN(lambda u, recv, send, nb, st: self._call(self.all_gather, recv, send, nb * world, nb))
        self._a2a = L.COMM_FN(lambda u, recv, send, nb, st: self._call(self.all_to_all, recv, send, nb * world, nb * world))

    def bh_comm(self):
        L = _lib()
        return L.BhComm(self.world, self.rank, None, self._ag, self._a2a, L.COMM_RELEASE_FN())

    def attach(self, stepper):
        self.tensors = [t for t in stepper._buffers if t is not None]
        self.stream = stepper.stream

    def _view(self, ptr, nbytes):
        for t in self.tensors:
            off = ptr - t.data_ptr()
            if 0 <= off and off + nbytes <= t.numel():
                return t[off:off + nbytes]
        raise RuntimeError("exchange pointer outside the registered buffers")

    def _call(self, fn, recv, send, recv_bytes, send_bytes):
        import contextlib
        try:
            with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
                fn(self._view(recv, recv_bytes), self._view(send, send_bytes))
            return 0
        except BaseException as ex:  # noqa: BLE001 - must not propagate through the C frames: BH_ERR_COMM, then
            self.error = ex          # DomainStepper.step re-raises it
            return 1

    def all_gather(self, out, send):
        raise NotImplementedError

    def all_to_all(self, out, send):
        raise NotImplementedError


class TorchComm(TensorComm):
    """the exchanges through torch.distributed ("nccl" = RCCL, or gloo: one-GPU rehearsals, CPU protocol tests)"""

    def __init__(self, group=None):
        super().__init__(dist.get_world_size(group), dist.get_rank(group))
        self.group = group
        self.into_tensor = _into_tensor_ok(group)   # chosen once, from the backend

    def all_gather(self, out, send):
        all_gather_rows(out.view(self.world, -1), send.view(1, -1), self.group, self.into_tensor)

    def all_to_all(self, out, send):
        """out chunk q <- rank q's send chunk `rank` (equal chunks)"""
        dist.all_to_all_single(out.view(-1), send.view(-1), group=self.group)


class RcclComm(_Comm):
    """the library's RCCL transport (bh_comm_rccl_init_rank): collectives on the rank's own stream, no torch in the
    step.  The 128-byte unique id comes from rank 0 through torch.distributed (any backend)."""

    def __init__(self, device, group=None):
        L = _lib()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        uid = (C.c_char * 128)()
        st = L.lib.bh_comm_rccl_unique_id(uid) if self.rank == 0 else 0
        box = [bytes(uid) if st == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            raise RuntimeError("RCCL is not available (bh_comm_rccl_unique_id failed on rank 0)")
        self._c = L.BhComm()
        st = L.lib.bh_comm_rccl_init_rank(C.byref(self._c), box[0], self.world, self.rank, int(device))
        if st != 0:
            raise RuntimeError(f"bh_comm_rccl_init_rank: {L.lib.bh_strerror(st).decode()}")

    def check(self):
        """bh_comm_check: one all-gather and one all-to-all of known words through this transport (collective)"""
        return _lib().lib.bh_comm_check(C.byref(self._c)) == 0

    def bh_comm(self):
        return self._c


class LocalGroup:
    """P ranks as P threads of one process on one device (tests, 1-GPU rehearsal): a bh_hub"""

    def __init__(self, world):
        import threading
        L = _lib()
        self.world = world
        self.barrier = threading.Barrier(world)
        self._h = C.c_void_p()
        st = L.lib.bh_hub_create(C.byref(self._h), world)
        if st != 0:
            raise RuntimeError(f"bh_hub_create: {st}")

    def abort(self):
        _lib().lib.bh_hub_abort(self._h)

    def __del__(self):
        try:
            if self._h:
                _lib().lib.bh_hub_destroy(self._h)
                self._h = None
        except Exception:
            pass


class LocalComm(_Comm):
    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def bh_comm(self):
        L = _lib()
        c = L.BhComm()
        st = L.lib.bh_comm_hub(C.byref(c), self.g._h, self.rank)
        if st != 0:
            raise RuntimeError(f"bh_comm_hub: {st}")
        return c


def global_morton_order(pkg, ic, device, params=None, **kw):
    """ids of the bodies in the key order of the global cube (one throw-away full-size engine:
    the same deterministic sort on every rank, so all ranks agree on the initial slabs)."""
    n = len(ic[0])
    with pkg.Engine(n, params=params, device=device, **kw) as e:
        e.upload(*ic)
        e.bbox()
        e.morton()
        e.sort()
        return e.download_order()


# Force passes per step when the caller does not choose: None = the library's rule (bh_rank_opts.split -1): ONE pass
# after X4.  split=True: the walk of the first split_pct (30) per cent of a rank's bodies in two passes — own pieces on a
# side stream behind the LET export, while X4 is in flight; remote pieces after X4, beside the one pass of the other
# bodies.  split="adaptive" (bh_rank_opts.split 2; bench.py): one pass while the measured X4 is short, the split form
# once it lasts ~0.2 ms and more — where it pays (profiles/r05_dd/replay_8x1M.txt, fake_x4_world1.txt).
SPLIT_DEFAULT = None

# X4 flavour of DomainStepper when the caller does not choose: per-destination segments + all-to-all
# (let_mode=0: the all-gather of one union segment per rank)
LET_MODE_DEFAULT = 1

BUFFER_NAMES = ("x1s", "x1r", "x2s", "x2r", "x3s", "x3r", "lets", "pool")


class DomainLeft(RuntimeError):
    """BH_ERR_DOMAIN_LEFT: raised by DomainStepper.step ON EVERY RANK after the same exchange (a rank-local failure
    announced through the X4 header, or a LET beyond let_cap decided from all-gathered counts): the only exception a
    caller may answer collectively, e.g. by switching every rank to another scheme.  Anything else is rank-local."""


class DomainStepper:
    """One rank of the domain-decomposed step — a binding of bh_rank: the per-step protocol (exchange order, size
    negotiation, retries, collective failure) runs inside the library (bh_rank_step, csrc/bh_group.hip)."""

    def __init__(self, pkg, ic, comm, device, stream=None, params=None, slack=1.3, mig_frac=0.5,
                 let_cap=None, order=None, split=None, let_mode=None, mig_log=False, split_pct=0, **kw):
        L = _lib()
        self.pkg, self.comm = pkg, comm
        if split is None:
            split = SPLIT_DEFAULT
        self.split = None if split is None else (2 if split == "adaptive" else int(bool(split)))
        self.let_mode = LET_MODE_DEFAULT if let_mode is None else int(let_mode)
        self.world, self.rank = comm.world, comm.rank
        P, r = self.world, self.rank
        n = len(ic[0])
        self.n_total = n
        if order is None:
            order = global_morton_order(pkg, ic, device, params=params, **kw)
        cuts = [q * n // P for q in range(P + 1)]
        mine = order[cuts[r]:cuts[r + 1]]
        x, y, z, vx, vy, vz, m = [a[mine] for a in ic]
        o = L.BhRankOpts()
        L.lib.bh_rank_default_opts(C.byref(o))
        if slack != 1.3 or mig_frac != 0.5:   # (the library's defaults are these)
            fair = max(cuts[q + 1] - cuts[q] for q in range(P))
            o.n_cap = max(1024, int(fair * slack) + 4096)
            o.mig_cap = min(max(4096, int(o.n_cap * mig_frac)), 4 * o.n_cap // P)
        if let_cap:
            o.let_cap = int(let_cap)
        o.let_mode, o.log = self.let_mode, int(bool(mig_log))
        o.split = -1 if self.split is None else int(self.split)
        o.split_pct = int(split_pct)
        o.serial = int(isinstance(comm, LocalComm))   # ranks of one process on one GPU: no side stream (bh_dd_set_serial)
        self.params = params if params is not None else pkg.default_params(**kw)
        plan = L.BhRankPlan()
        st = L.lib.bh_rank_query(n, P, C.byref(o), C.byref(plan))
        if st != 0:
            raise pkg.BhError(st, "bh_rank_query")
        if self.split is None:   # (bh_rank_opts.split -1: one pass)
            self.split = 0
        self.stream = stream if stream is not None else torch.cuda.Stream(device)
        bufs = None
        self._buffers = []
        if isinstance(comm, TensorComm):   # the transport moves torch tensors: the buffers are ours
            with torch.cuda.stream(self.stream):
                self._buffers = [torch.zeros(int(plan.bytes[k]), dtype=torch.uint8, device=f"cuda:{device}")
                                 for k in range(8)]
            self.stream.synchronize()
            bufs = L.BhRankBuffers(*[t.data_ptr() for t in self._buffers])
        self._c = comm.bh_comm()
        self._h = C.c_void_p()
        st = L.lib.bh_rank_create(C.byref(self._h), C.byref(self._c), n, C.byref(self.params), C.byref(o), int(device),
                                  C.c_void_p(self.stream.cuda_stream), C.byref(bufs) if bufs is not None else None)
        if st != 0:
            self._h = C.c_void_p()
            raise pkg.BhError(st, "bh_rank_create")
        self._finish(plan, mig_log)
        comm.attach(self)
        ids = np.ascontiguousarray(mine, dtype=np.int32)
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, z, vx, vy, vz, m)]
        st = L.lib.bh_rank_upload(self._h, len(ids), *[a.ctypes.data_as(L._F) for a in arrs],
                                  ids.ctypes.data_as(C.POINTER(C.c_int32)))
        if st != 0:
            raise pkg.BhError(st, "bh_rank_upload")
        self.e.n = len(ids)

    def _finish(self, plan, mig_log):
        L = _lib()
        self.plan, self.sz = plan, plan.sz
        self.n_cap, self.mig_cap, self.let_cap = plan.n_cap, plan.mig_cap, plan.let_cap
        self._log = bool(mig_log)
        h = L.lib.bh_rank_ctx(self._h)
        self.e = self.pkg.Engine.adopt(h, plan.n_cap, getattr(self, "params", None)) if h else None
        self._sync_info()

    @classmethod
    def with_engine(cls, engine, sz, comm, n_cap, mig_cap, let_cap, tensor_device="cpu", split=True, let_mode=0):
        """The library's per-step protocol (bh_rank_create_scripted) around ANY object with the dd_* methods of
        Engine, on host buffers — tests/dd_cpu_worker.py drives it on CPU tensors over gloo."""
        import bhpkg
        L = _lib()
        self = cls.__new__(cls)
        self.pkg = bhpkg.load()
        self.comm, self.split, self.let_mode = comm, bool(split), int(let_mode)
        self.world, self.rank = comm.world, comm.rank
        self.stream = None
        P = self.world
        plan = L.BhRankPlan()
        plan.n_cap, plan.mig_cap, plan.let_cap = int(n_cap), int(mig_cap), int(let_cap)
        plan.stride0 = min(int(let_cap), _round_up(sz.let_min + int(n_cap) // 8, 256))
        for k in ("x1_bytes", "x2_bytes", "x3_bytes", "pool_records", "seg_base", "let_min", "let_cap", "top_base"):
            setattr(plan.sz, k, int(getattr(sz, k)))
        nseg = P if self.let_mode == 1 else 1
        for k, b in enumerate((sz.x1_bytes, P * sz.x1_bytes, sz.x2_bytes, P * sz.x2_bytes, sz.x3_bytes,
                               P * sz.x3_bytes, nseg * int(let_cap) * 32, sz.pool_records * 32)):
            plan.bytes[k] = int(b)
        o = L.BhRankOpts()
        L.lib.bh_rank_default_opts(C.byref(o))
        o.let_mode, o.split = self.let_mode, int(self.split)
        self._script_error = None

        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as ex:  # noqa: BLE001 - reported as a status, re-raised by step()
                    self._script_error = ex
                    return -5
            return run

        def phase_migrate(u, x1r, x2s, limit):
            engine.dd_cube_apply(x1r)
            engine.dd_migrate_pack(x2s, limit)

        def phase_tree(u, x2r, limit, x3s, n_loc, more, most):
            a, b, c = engine.dd_migrate_apply(x2r, limit)
            n_loc[0], more[0], most[0] = int(a), int(bool(b)), int(c)
            if not b:
                engine.dd_tree(x3s)

        def phase_let(u, x3r, x4s, stride, own):
            if own:
                engine.dd_force_local(x3r)
            engine.dd_let_pack(x3r, x4s, stride)

        def phase_force(u, x3r, stride, counts, fits):
            engine.dd_top(x3r, stride)
            ok, cnt = engine.dd_let_check(stride, P)
            for q in range(P):
                counts[q] = int(cnt[q])
            fits[0] = int(bool(ok))
            if ok and int(min(cnt)) >= 0:
                engine.dd_force()

        def phase_end(u, x1s):
            engine.integrate()
            engine.dd_cube_pack(x1s)

        fns = dict(cube_pack=lambda u, p: engine.dd_cube_pack(p), phase_migrate=phase_migrate,
                   migrate_pack=lambda u, p, limit: engine.dd_migrate_pack(p, limit), phase_tree=phase_tree,
                   phase_let=phase_let, phase_force=phase_force, phase_end=phase_end)
        self._script = L.BhRankScript()
        self._script_keep = []
        for name, ftype in L.SCRIPT_FNS:
            cb = ftype(guard(fns[name]))
            self._script_keep.append(cb)
            setattr(self._script, name, cb)
        self._c = comm.bh_comm()
        self._h = C.c_void_p()
        st = L.lib.bh_rank_create_scripted(C.byref(self._h), C.byref(self._c), C.byref(self._script), C.byref(plan),
                                           C.byref(o))
        if st != 0:
            raise self.pkg.BhError(st, "bh_rank_create_scripted")
        b = L.BhRankBuffers()
        L.lib.bh_rank_buffers_of(self._h, C.byref(b), None)
        self._buffers = [torch.frombuffer((C.c_char * int(plan.bytes[k])).from_address(getattr(b, f)), dtype=torch.uint8)
                         for k, f in enumerate(("x1s", "x1r", "x2s", "x2r", "x3s", "x3r", "x4s", "pool"))]
        self.e = engine
        self.plan, self.sz = plan, sz
        self.n_cap, self.mig_cap, self.let_cap = int(n_cap), int(mig_cap), int(let_cap)
        self._log = False
        self._sync_info()
        comm.attach(self)
        return self

    def __getattr__(self, name):   # x1s ... pool: the exchange buffers as tensors (scripted ranks, TorchComm)
        if name in BUFFER_NAMES and self.__dict__.get("_buffers"):
            return self._buffers[BUFFER_NAMES.index(name)]
        raise AttributeError(name)

    def _sync_info(self):
        L = _lib()
        i = L.BhRankInfo()
        L.lib.bh_rank_get_info(self._h, C.byref(i))
        self.n_loc, self.stride, self.mig_stride, self.mig_last = i.n_loc, i.stride, i.mig_stride, i.mig_last
        self.mig_rounds, self.let_retries = i.mig_rounds, i.let_retries
        self.split_now, self.x4_us = int(i.split_now), int(i.x4_us)   # (adaptive form: what it runs now, measured X4)
        self.x4_recv_bytes = int(i.x4_recv_kb) * 1024   # what this rank received in the last X4
        self.let_counts = np.array(i.let_counts[:self.world], np.int32) if i.steps or i.let_retries else None
        self._info = i
        if getattr(self, "e", None) is not None and hasattr(self.e, "_h") and i.n_loc:
            self.e.n = i.n_loc

    # ---- optional per-phase device timing (events on the rank's stream; bench.py reports the means) ----
    PHASES = ("x1_exchange", "cube_splitters_x2_migration_local_tree", "x3_exchange", "let_export_x4",
              "top_remote_force", "integrate_pack_x1")

    def set_profile(self, on=True):
        _lib().lib.bh_rank_set_profile(self._h, 1 if on else 0)
        self._profiled = bool(on)

    def phase_ms(self):
        """mean device time per phase over the profiled steps"""
        if not getattr(self, "_profiled", False):
            return None
        ms = (C.c_double * 6)()
        cnt = C.c_int(0)
        _lib().lib.bh_rank_phase_ms(self._h, ms, C.byref(cnt))
        if cnt.value == 0:
            return None
        return {name: float(ms[k]) for k, name in enumerate(self.PHASES)}

    @property
    def mig_log(self):
        """per step (most emigrants on any rank, what the step did with the boundaries) — mig_log=True only"""
        if not self._log:
            return None
        L = _lib()
        n = C.c_int(0)
        L.lib.bh_rank_read_log(self._h, None, 0, C.byref(n))
        buf = np.zeros((max(n.value, 1), 2), np.int32)
        L.lib.bh_rank_read_log(self._h, buf.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n))
        return [(int(a), int(b)) for a, b in buf[:n.value]]

    def step(self, steps=1):
        """bh_rank_step: `steps` steps of  [X1] phase_migrate [X2] phase_tree [X3] phase_let [X4] phase_force,
        phase_end  (collective: every rank calls it with the same count)"""
        L = _lib()
        st = L.lib.bh_rank_step(self._h, int(steps))
        self._sync_info()
        if st == 0:
            return
        i = self._info
        if st == L.BH_ERR_DOMAIN_LEFT:
            if i.left_rank == self.rank:
                why = getattr(self, "_script_error", None)
                why = repr(why) if why is not None else self.pkg.lib.bh_strerror(i.left_status).decode()
                raise DomainLeft(f"rank {self.rank} left the domain-decomposed step: {why}")
            if i.left_rank >= 0:
                raise DomainLeft(f"rank {i.left_rank} left the domain-decomposed step")
            raise DomainLeft(f"LET of {int(max(i.let_counts[:self.world]))} records exceeds let_cap {self.let_cap}; "
                             "every rank left the domain-decomposed step")
        err = getattr(self.comm, "error", None) or getattr(self, "_script_error", None)
        if err is not None:
            raise err
        raise self.pkg.BhError(st, "bh_rank_step")

    def local_state(self):
        """(ids, posm[n,4], vel[n,3], acc[n,3]) of this rank's bodies"""
        posm, vel, ids, acc = self.e.dd_download()
        return ids, posm, vel, acc

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib().lib.bh_rank_destroy(self._h)
            self._h = None
            if hasattr(self.e, "_h"):
                self.e._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
