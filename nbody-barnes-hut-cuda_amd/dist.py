"""Multi-GPU stepping: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-GPU (no NCCL/MPI call sites); this layer is new design
(SURVEY §8e).  Round-1 scheme — exact by construction:

  * every rank holds the full particle state and builds the SAME tree (bbox, keys, sort, build
    and COM are deterministic, so the replicas stay bit-identical without any exchange);
  * the force stage — >90 % of the step — is sharded: rank r traverses only the Morton slab
    [r*slab, (r+1)*slab) of the sorted bodies (a contiguous spatial domain);
  * ONE collective per step: all-gather of the per-rank acceleration slabs (float4 per body)
    straight into the engine's acceleration buffer; every rank then integrates all bodies.

P-rank results are therefore bit-identical to 1-rank results.  The replicated build is the
Amdahl term; DESIGN.md ("what comes next") describes the domain-decomposed build + top-tree /
LET all-gather that replaces it.
"""
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


# ----------------------------------------------------------------------------------------------
# Domain-decomposed stepping (include/bh.h bh_dd_*, csrc/bh_dd.hip)
# ----------------------------------------------------------------------------------------------
class TorchComm:
    """the four per-step all-gathers over torch.distributed (backend "nccl" = RCCL over xGMI)"""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.into_tensor = _into_tensor_ok(group)   # chosen once, from the backend

    def all_gather(self, out, send):
        all_gather_rows(out.view(self.world, -1), send.view(1, -1), self.group, self.into_tensor)

    def all_to_all(self, out, send):
        """out chunk q <- rank q's send chunk `rank` (equal chunks)"""
        dist.all_to_all_single(out.view(-1), send.view(-1), group=self.group)


class LocalGroup:
    """P ranks as P threads of one process on one device and ONE stream (tests, 1-GPU rehearsal):
    the all-gather is P device copies; a barrier orders the ranks' enqueues on the shared stream."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class LocalComm:
    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def all_gather(self, out, send):
        g = self.g
        g.slots[self.rank] = send
        g.barrier.wait()
        n = send.numel()
        o = out.view(-1)
        for q in range(self.world):
            o[q * n:(q + 1) * n].copy_(g.slots[q].view(-1))
        g.barrier.wait()

    def all_to_all(self, out, send):
        g = self.g
        g.slots[self.rank] = send
        g.barrier.wait()
        k = send.numel() // self.world
        o = out.view(-1)
        for q in range(self.world):
            o[q * k:(q + 1) * k].copy_(g.slots[q].view(-1)[self.rank * k:(self.rank + 1) * k])
        g.barrier.wait()


def _round_up(v, a):
    return (int(v) + a - 1) // a * a


def global_morton_order(pkg, ic, device, params=None, **kw):
    """ids of the bodies in the Morton order of the global cube (one throw-away full-size engine:
    the same deterministic sort on every rank, so all ranks agree on the initial slabs)."""
    n = len(ic[0])
    with pkg.Engine(n, params=params, device=device, **kw) as e:
        e.upload(*ic)
        e.bbox()
        e.morton()
        e.sort()
        return e.download_order()


# One force pass per step when the caller does not choose (split=False): the LET export and X4 are exposed, but the
# two-pass form — own pieces on a side stream while the LET travels, then the remote pass — costs 0.42 ms more GPU
# time per rank-step at 8 x 1M (own 1.15-1.25 + remote 0.50 against 1.23 ms for the one pass: two drains, the top
# levels walked twice, profiles/r04_dd/split_vs_one_pass.txt), more than the ~0.3 ms (0.125 ms of LET kernels + a
# 33-MB all-to-all over xGMI) it can hide.  split=True remains for interconnects slow enough to turn that around.
SPLIT_DEFAULT = False

# X4 flavour of DomainStepper when the caller does not choose: per-destination segments + all-to-all
# (DomainStepper(..., let_mode=0) selects round 2's all-gather of the union; tools/dd_debug.py --let-mode 0 for A/B)
LET_MODE_DEFAULT = 1


class DomainLeft(RuntimeError):
    """Raised by DomainStepper.step ON EVERY RANK after the same exchange (a rank-local failure announced through
    the X4 header, or a LET beyond let_cap decided from all-gathered counts): the only exception a caller may
    answer collectively, e.g. by switching every rank to another scheme.  Anything else is rank-local."""


class DomainStepper:
    """One rank of the domain-decomposed step: this rank owns the bodies of one Morton-key range,
    builds only their octree, imports the other ranks' locally-essential records and traverses the
    stitched tree for its own bodies.  Four all-gathers per step, no replicated stage."""

    def __init__(self, pkg, ic, comm, device, stream=None, params=None, slack=1.3, mig_frac=0.5,
                 let_cap=None, order=None, split=None, let_mode=None, mig_log=False, **kw):
        self.comm = comm
        self.split = SPLIT_DEFAULT if split is None else bool(split)
        # X4: 0 = all-gather of the union segment (round 2), 1 = per-destination segments, all-to-all
        self.let_mode = LET_MODE_DEFAULT if let_mode is None else int(let_mode)
        self.world, self.rank = comm.world, comm.rank
        P, r = self.world, self.rank
        n = len(ic[0])
        self.n_total = n
        if order is None:
            order = global_morton_order(pkg, ic, device, params=params, **kw)
        cuts = [q * n // P for q in range(P + 1)]
        counts = [cuts[q + 1] - cuts[q] for q in range(P)]
        mine = order[cuts[r]:cuts[r + 1]]
        x, y, z, vx, vy, vz, m = [a[mine] for a in ic]

        self.n_cap = max(1024, int(max(counts) * slack) + 4096)
        # X2 buffers hold up to mig_cap emigrants per rank; a step normally sends far fewer
        # (mig_stride follows the observed count), a larger wave goes in several rounds
        self.mig_cap = min(max(4096, int(self.n_cap * mig_frac)), 4 * self.n_cap // P)
        self.mig_stride = min(self.mig_cap, 4096)
        self.mig_rounds = 0
        self.mig_last = 0
        self.mig_log = [] if mig_log else None   # per step (emigrants, boundaries): synchronises — tests and tools only
        e_cls = pkg.Engine
        lmin = 4 + 512  # header + needs row + piece slots (BH_DD_PIECE_CAP, csrc/bh_dd.hip kSegBlocks0)
        self.let_cap = int(let_cap) if let_cap else lmin + self.n_cap
        self.let_cap += self.let_cap & 1                        # segments hold whole 64-byte digest pairs
        sz = e_cls.dd_query(self.n_cap, P, self.mig_cap, self.let_cap)
        self.sz = sz
        self.stream = stream if stream is not None else torch.cuda.Stream(device)
        dev = f"cuda:{device}"
        with torch.cuda.stream(self.stream):
            u8 = dict(dtype=torch.uint8, device=dev)
            self.x1s = torch.zeros(sz.x1_bytes, **u8)
            self.x1r = torch.zeros(P * sz.x1_bytes, **u8)
            self.x2s = torch.zeros(sz.x2_bytes, **u8)
            self.x2r = torch.zeros(P * sz.x2_bytes, **u8)
            self.x3s = torch.zeros(sz.x3_bytes, **u8)
            self.x3r = torch.zeros(P * sz.x3_bytes, **u8)
            self.lets = torch.zeros((P if self.let_mode == 1 else 1) * self.let_cap * 32, **u8)
            self.pool = torch.zeros(sz.pool_records * 32, **u8)
        self.stream.synchronize()
        assert sz.let_min == lmin, (sz.let_min, lmin)
        self.e = e_cls(self.n_cap, params=params, device=device, stream=self.stream.cuda_stream, **kw)
        self.e.dd_init(P, r, n, self.mig_cap, self.let_cap, self.pool.data_ptr(), sz.pool_records)
        self.e.dd_set_let_mode(self.let_mode)
        self.e.dd_upload(x, y, z, vx, vy, vz, m, mine.astype("int32"))
        self.stride = min(self.let_cap, _round_up(lmin + self.n_cap // 8, 256))
        self.let_counts = None
        self.let_retries = 0
        self.n_loc = len(mine)

    @classmethod
    def with_engine(cls, engine, sz, comm, n_cap, mig_cap, let_cap, tensor_device="cpu", split=True, let_mode=0):
        """The per-step protocol (exchanges, size negotiation, failure handling) around ANY object with
        the dd_* methods of Engine — tests/dd_cpu_worker.py drives it on CPU tensors over gloo."""
        self = cls.__new__(cls)
        self.comm, self.split = comm, bool(split)
        self.let_mode = int(let_mode)
        self.world, self.rank = comm.world, comm.rank
        self.e, self.sz, self.stream = engine, sz, None
        self.n_cap, self.mig_cap, self.let_cap = int(n_cap), int(mig_cap), int(let_cap)
        self.mig_stride = min(self.mig_cap, 4096)
        self.mig_rounds = self.mig_last = self.let_retries = self.n_loc = 0
        self.mig_log = None
        self.let_counts = None
        P = self.world
        u8 = dict(dtype=torch.uint8, device=tensor_device)
        self.x1s = torch.zeros(sz.x1_bytes, **u8)
        self.x1r = torch.zeros(P * sz.x1_bytes, **u8)
        self.x2s = torch.zeros(sz.x2_bytes, **u8)
        self.x2r = torch.zeros(P * sz.x2_bytes, **u8)
        self.x3s = torch.zeros(sz.x3_bytes, **u8)
        self.x3r = torch.zeros(P * sz.x3_bytes, **u8)
        self.lets = torch.zeros((P if self.let_mode == 1 else 1) * self.let_cap * 32, **u8)
        self.pool = torch.zeros(sz.pool_records * 32, **u8)
        self.stride = min(self.let_cap, _round_up(sz.let_min + self.n_cap // 8, 256))
        return self

    # ---- optional per-phase device timing (events on the main stream; bench.py reports the means) ----
    PHASES = ("x1_exchange", "cube_splitters_x2_migration_local_tree", "x3_exchange", "let_export_x4",
              "top_remote_force", "integrate_pack_x1")

    def set_profile(self, on=True):
        self._prof = [] if on and self.stream is not None else None

    def _mark(self, k):
        if getattr(self, "_prof", None) is None:
            return
        if k == 0:
            self._cur = [None] * 7
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.stream)
        self._cur[k] = ev          # a repeated phase (LET retry) keeps its last mark

    def _close_marks(self):
        if getattr(self, "_prof", None) is not None:
            self._prof.append(self._cur)

    def phase_ms(self):
        """mean device time per phase over the profiled steps (the own force pass overlaps the LET phase
        on its own stream and is waited for inside 'top_remote_force')"""
        if not getattr(self, "_prof", None):
            return None
        torch.cuda.synchronize()
        acc = [0.0] * 6
        for evs in self._prof:
            for k in range(6):
                acc[k] += evs[k].elapsed_time(evs[k + 1])
        return {name: acc[k] / len(self._prof) for k, name in enumerate(self.PHASES)}

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    # ---- phase groups: one C call each (bh_dd_phase_*) when the engine has them; the scripted stand-in of
    # tests/dd_cpu_worker.py only has the fine-grained calls, which these fall back to
    def _phase_migrate(self, limit):
        e = self.e
        if hasattr(e, "dd_phase_migrate"):
            e.dd_phase_migrate(self.x1r.data_ptr(), self.x2s.data_ptr(), limit)
        else:
            e.dd_cube_apply(self.x1r.data_ptr())
            e.dd_migrate_pack(self.x2s.data_ptr(), limit)

    def _phase_tree(self, limit):
        e = self.e
        if hasattr(e, "dd_phase_tree"):
            return e.dd_phase_tree(self.x2r.data_ptr(), limit, self.x3s.data_ptr())
        n_loc, more, most = e.dd_migrate_apply(self.x2r.data_ptr(), limit)
        if not more:
            e.dd_tree(self.x3s.data_ptr())
        return n_loc, more, most

    def _phase_let(self, stride, own_pass):
        e = self.e
        if hasattr(e, "dd_phase_let"):
            e.dd_phase_let(self.x3r.data_ptr(), self.lets.data_ptr(), stride, own_pass)
        else:
            if own_pass:
                e.dd_force_local(self.x3r.data_ptr())
            e.dd_let_pack(self.x3r.data_ptr(), self.lets.data_ptr(), stride)

    def _phase_force(self, stride):
        e = self.e
        if hasattr(e, "dd_phase_force"):
            return e.dd_phase_force(self.x3r.data_ptr(), stride, self.world)
        e.dd_top(self.x3r.data_ptr(), stride)
        e.dd_force()
        return e.dd_let_check(stride, self.world)

    def _phase_end(self):
        e = self.e
        if hasattr(e, "dd_phase_end"):
            e.dd_phase_end(self.x1s.data_ptr())
        else:
            e.integrate()
            e.dd_cube_pack(self.x1s.data_ptr())
        self._x1_ready = True   # the next step's X1 payload is packed

    def step(self, steps=1):
        """Five library calls and four all-gathers per step in the common case:
             [X1] phase_migrate [X2] phase_tree [X3] phase_let [X4] phase_force, phase_end
           (extra migration rounds, a LET retry and the collective handling of a rank-local failure use the
           fine-grained entry points)."""
        e, c, sz, P = self.e, self.comm, self.sz, self.world
        with self._on_stream():
            for _ in range(int(steps)):
                self._mark(0)
                if not getattr(self, "_x1_ready", False):
                    e.dd_cube_pack(self.x1s.data_ptr())                # X1: cube + splitters (first step only:
                self._x1_ready = False                                 # afterwards the previous step packed it)
                c.all_gather(self.x1r, self.x1s)
                self._mark(1)
                limit, first = self.mig_stride, None                   # X2: bodies that changed owner
                self.mig_stride_used = limit                           # (slots per rank of this step's first round)
                failed = None   # a rank-local failure must not strand the others inside a collective: the
                rounds = 0      # failing rank keeps taking part with empty payloads and marks its LET segment
                while True:
                    nb = 32 + 32 * limit
                    if failed is not None:
                        self.x2s[:32].zero_()
                    elif rounds == 0:
                        self._phase_migrate(limit)                     # global cube + splitters, emigrants packed
                    else:
                        e.dd_migrate_pack(self.x2s.data_ptr(), limit)
                    rounds += 1
                    c.all_gather(self.x2r[:P * nb], self.x2s[:nb])
                    if failed is None:
                        try:   # immigrants absorbed; once no rank has emigrants left: local sort / build / COM,
                            self.n_loc, more, most = self._phase_tree(limit)   # X3 piece descriptors packed
                        except Exception as ex:  # noqa: BLE001 - e.g. more bodies than this context can hold
                            failed = ex
                    if failed is not None:
                        h = self.x2r[:P * nb].view(P, nb)[:, :16].contiguous().view(torch.int32).cpu().view(P, 4)
                        more, most = bool(((h[:, 0] - h[:, 2]) > 0).any()), int(h[:, 0].max())
                    first = most if first is None else first
                    if not more:
                        break
                    self.mig_rounds += 1                                # rare: a splitter changed octant
                    limit = min(self.mig_cap, max(limit, _round_up(most, 256)))
                self.mig_last = first
                if self.mig_log is not None and hasattr(e, "dd_info"):   # tests / tools: (emigrants, boundaries) per step
                    self.mig_log.append((first, e.dd_info()[3]))
                # next step's slots: one and a half times what this step moved (boundaries that persist move a fraction of a per
                # cent of a rank per step; a rebalance or a collapse moves more and goes in several rounds)
                self.mig_stride = max(1024, min(self.mig_cap, _round_up(first * 1.5 + 512, 256)))
                self._mark(2)
                if failed is not None:
                    self.x3s.zero_()
                c.all_gather(self.x3r, self.x3s)
                self._mark(3)
                tries = 0
                while True:
                    stride = self.stride
                    seg = self.pool[sz.seg_base * 32:(sz.seg_base + P * stride) * 32]
                    nseg = P if self.let_mode == 1 else 1               # segments this rank sends
                    send = self.lets[:nseg * stride * 32]
                    x4 = c.all_to_all if self.let_mode == 1 else c.all_gather
                    if failed is not None:
                        send.zero_()
                        # header count < 0: "this rank failed" (record 0 of a digest pair: field `first` is dword 10,
                        # csrc/bh_internal.h) — in every segment it sends
                        send.view(nseg, stride * 32)[:, :64].view(torch.int32)[:, 10] = -1
                        x4(seg, send)
                        raise DomainLeft(f"rank {self.rank} left the domain-decomposed step: {failed!r}")
                    # own pieces on the side stream (first try only: it overlaps X4), LET marked / exported
                    self._phase_let(stride, self.split and tries == 0)
                    tries += 1
                    x4(seg, send)                                       # X4: LET records, in place
                    self._mark(4)
                    ok, counts = self._phase_force(stride)              # top tree, remote (or whole) pass, X4 sizes
                    self.let_counts = counts
                    if int(counts.min()) < 0:
                        raise DomainLeft(f"rank {int(counts.argmin())} left the domain-decomposed step")
                    need = int(counts.max())
                    if ok:
                        break
                    if need > self.let_cap:
                        raise DomainLeft(f"LET of {need} records exceeds let_cap {self.let_cap}")
                    self.let_retries += 1
                    self.stride = min(self.let_cap, _round_up(need * 1.25, 256))
                # every rank sees the same counts, so every rank picks the same next stride
                self.stride = max(sz.let_min, min(self.let_cap, _round_up(need * 1.15 + 1024, 256)))
                self._mark(5)
                self._phase_end()                                       # integrate + the next step's X1 payload
                self._mark(6)
                self._close_marks()

    def local_state(self):
        """(ids, posm[n,4], vel[n,3], acc[n,3]) of this rank's bodies"""
        posm, vel, ids, acc = self.e.dd_download()
        return ids, posm, vel, acc

    def close(self):
        self.e.close()
