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


def all_gather_rows(out, send, group=None):
    """out[world*slab, 4] <- concatenation of every rank's send[slab, 4]."""
    try:
        dist.all_gather_into_tensor(out, send, group=group)
    except (RuntimeError, NotImplementedError):
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
        assert acc.shape == (self.world * self.slab, 4) and acc.dtype == torch.float32
        self.send = torch.zeros((self.slab, 4), dtype=acc.dtype, device=acc.device)

    def step(self, steps=1):
        for _ in range(int(steps)):
            self.e.tree_stages()                 # replicated, deterministic
            self.e.force(self.lo, self.hi)       # sharded
            if self.world > 1:                   # the one exchange step
                row0 = self.rank * self.slab
                self.send.copy_(self.acc[row0:row0 + self.slab])
                all_gather_rows(self.acc, self.send, self.group)
            self.e.integrate()                   # replicated


def make_gpu_stepper(pkg, n, params=None, device=None, group=None, **kw):
    """Engine on torch's current stream (so RCCL collectives and kernels are ordered by the
    stream) with its acceleration buffer bound to a torch tensor sized for the all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if device is None:
        device = torch.cuda.current_device()
    slab = slab_size(n, world)
    acc = torch.zeros((world * slab, 4), dtype=torch.float32, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(device).cuda_stream
    eng = pkg.Engine(n, params=params, device=device, stream=stream, **kw)
    eng.bind_acc(acc.data_ptr())
    eng._acc_keepalive = acc
    return eng, ShardedStepper(eng, acc, n, group)
