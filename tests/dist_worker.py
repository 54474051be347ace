"""Worker for tests/test_dist_cpu.py — run under torch.distributed.run with the gloo backend.

Drives the PRODUCT's multi-rank stepping logic (nbody_barnes_hut_cuda_amd.dist.ShardedStepper:
slab bounds, the one all-gather per step, replicated integrate) with an oracle-backed CPU stand-in
for the GPU engine, so the N > 1 path is exercised without a GPU.  Rank 0 writes the final state.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bhpkg  # noqa: E402
import oracle as O  # noqa: E402


class OracleEngine:
    """Engine-like object (tree_stages / force(lo,hi) / integrate) on CPU tensors."""

    def __init__(self, ic, p, acc):
        x, y, z, vx, vy, vz, m = ic
        self.n = len(x)
        self.p = p
        self.xyzm = np.stack([x, y, z, m], 1).astype(np.float32)
        self.vel = np.stack([vx, vy, vz], 1).astype(np.float32)
        self.ids = np.arange(self.n, dtype=np.int32)
        self.acc = acc  # torch [world*slab, 4], shared with the stepper
        self.rec = None

    def tree_stages(self):
        b = O.bbox(self.xyzm[:, 0], self.xyzm[:, 1], self.xyzm[:, 2])
        k = O.keys(self.xyzm[:, 0], self.xyzm[:, 1], self.xyzm[:, 2], b, self.p.key_bits, self.p.key_curve)
        sk, perm = O.sort(k)
        self.xyzm = np.ascontiguousarray(self.xyzm[perm])
        self.vel = np.ascontiguousarray(self.vel[perm])
        self.ids = self.ids[perm]
        rec, lo, hi, _, _ = O.build_tree(sk, self.p, O.root_edge(b))
        self.rec = O.com(rec, lo, hi, self.xyzm)

    def force(self, lo, hi):
        a, *_ = O.force(self.rec, self.xyzm, self.p, O.ORDER_PREORDER, lo, hi, counters=False, nthreads=2)
        self.acc[lo:hi] = torch.from_numpy(a[lo:hi])

    def integrate(self):
        a = self.acc[: self.n].numpy()
        self.xyzm, self.vel = O.integrate(self.xyzm, self.vel, a, self.p)

    def state_caller_order(self):
        out = np.zeros((self.n, 6), np.float32)
        out[self.ids, :3] = self.xyzm[:, :3]
        out[self.ids, 3:] = self.vel
        return out


def main():
    out_path, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    pkg = bhpkg.load()
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    world, rank = dist.get_world_size(), dist.get_rank()
    ic = pkg.plummer(n, seed=42)
    p = O.params()
    slab = bhdist.slab_size(n, world)
    acc = torch.zeros((world * slab, 4), dtype=torch.float32)
    eng = OracleEngine(ic, p, acc)
    st = bhdist.ShardedStepper(eng, acc, n)
    assert st.world == world and st.rank == rank
    st.step(steps)
    state = eng.state_caller_order()
    # replicas must stay bit-identical: compare every rank's state with rank 0's
    t = torch.from_numpy(state.copy())
    dist.broadcast(t, src=0)
    same = torch.tensor([int(np.array_equal(t.numpy(), state))])
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        np.savez(out_path, state=state, replicas_identical=int(same.item()), world=world,
                 slab=slab, lo=st.lo, hi=st.hi)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
