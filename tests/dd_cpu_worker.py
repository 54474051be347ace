"""Worker for tests/test_dist_cpu.py::test_domain_stepper_protocol — gloo, CPU tensors, no GPU.

Drives the PRODUCT's per-step protocol of the domain-decomposed multi-GPU step
(nbody_barnes_hut_cuda_amd.dist.DomainStepper.step: the four exchanges, the adaptive X2 / X4 sizes,
extra migration rounds, the LET retry, the collective handling of a rank-local failure) with a
scripted stand-in for the engine that only writes and reads the buffer HEADERS the protocol looks at.
The physics of the scheme is covered on the GPU (tests/test_gpu_dd.py, tests/test_gpu_dist.py)."""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402

PIECES = 512


class ScriptedEngine:
    """emigrants[step] / let_need[step] are per-rank lists; fail_at = (step, rank) raises in migrate_apply"""

    def __init__(self, rank, world, emigrants, let_need, fail_at=None):
        self.rank, self.world = rank, world
        self.emig, self.need, self.fail_at = emigrants, let_need, fail_at
        self.step_no = -1
        self.left = 0
        self.calls = []

    def bind(self, st):
        self.st = st

    # X1
    def dd_cube_pack(self, ptr):   # the product packs the NEXT step's X1 payload at the end of every step
        self.calls.append("cube_pack")
        if self.step_no + 1 >= len(self.emig):
            return                 # the pack after the last scripted step
        self.step_no += 1
        self.left = self.emig[self.step_no][self.rank]

    def dd_cube_apply(self, ptr):
        pass

    # X2: header ints [found, kept, sent, held]
    def dd_migrate_pack(self, ptr, limit):
        sent = min(self.left, limit)
        h = self.st.x2s[:16].view(torch.int32)
        h[0], h[1], h[2], h[3] = self.left, 1000, sent, 1000 + self.left - sent
        self.left -= sent
        self.calls.append(("migrate_pack", limit))

    def dd_migrate_apply(self, ptr, limit):
        if self.fail_at == (self.step_no, self.rank):
            raise RuntimeError("scripted capacity failure")
        P, nb = self.world, 32 + 32 * limit
        h = self.st.x2r[:P * nb].view(P, nb)[:, :16].contiguous().view(torch.int32).view(P, 4)
        more = bool(((h[:, 0] - h[:, 2]) > 0).any())
        return 1000, more, int(h[:, 0].max())

    def dd_tree(self, ptr):
        self.st.x3s[:8].view(torch.int32)[0] = 3  # three pieces

    def dd_force_local(self, ptr):
        self.calls.append("force_local")

    # X4: record 0 of the segment (slot 0 of a 64-byte digest pair), field `first` = dword 10 = records needed
    def dd_let_pack(self, x3ptr, sendptr, stride):
        # all-gather flavour: one segment; per-destination flavour: one segment per receiver, each with the header
        nseg = self.world if self.st.let_mode == 1 else 1
        need = self.need[self.step_no][self.rank]
        self.st.lets[:nseg * stride * 32].view(nseg, stride * 32)[:, :64].view(torch.int32)[:, 10] = need
        self.calls.append(("let_pack", stride))

    def dd_top(self, x3ptr, stride):
        pass

    def dd_force(self):
        pass

    def dd_let_check(self, stride, P):
        sz = self.st.sz
        seg = self.st.pool[sz.seg_base * 32:(sz.seg_base + P * stride) * 32].view(P, stride * 32)
        counts = np.array([int(seg[q, :64].view(torch.int32)[10]) for q in range(P)], np.int32)
        return bool(counts.max() <= stride), counts

    def integrate(self):
        self.calls.append("integrate")


def main():
    out_path, scenario = sys.argv[1], sys.argv[2]
    let_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dist.init_process_group("gloo")
    bhpkg.load()
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    comm = bhdist.TorchComm()
    P, r = comm.world, comm.rank
    n_cap, mig_cap, let_cap = 20000, 10000, 4 + PIECES + 20000
    sz = types.SimpleNamespace(x1_bytes=64, x2_bytes=32 + 32 * mig_cap, x3_bytes=80 * (1 + PIECES),
                               pool_records=2 * n_cap + 8 + 100 + P * let_cap + 8, seg_base=2 * n_cap + 108,
                               let_min=4 + PIECES, let_cap=let_cap, top_base=2 * n_cap + 8)
    steps = 4
    emig = [[10, 20, 5][:P] + [0] * (P - 3), [6000] * P, [9000] + [100] * (P - 1), [50] * P]
    need = [[600] * P, [700 + 4000 * (q == 1) for q in range(P)], [15000] * P, [15500] * P]
    fail_at = (2, P - 1) if scenario == "failure" else None
    eng = ScriptedEngine(r, P, emig, need, fail_at)
    st = bhdist.DomainStepper.with_engine(eng, sz, comm, n_cap, mig_cap, let_cap, let_mode=let_mode)
    eng.bind(st)
    err = None
    try:
        st.step(steps)
    except RuntimeError as ex:
        err = str(ex)
    res = {"rank": r, "error": err, "mig_rounds": st.mig_rounds, "let_retries": st.let_retries,
           "stride": int(st.stride), "mig_stride": int(st.mig_stride), "steps_done": eng.calls.count("integrate"),
           "integrates": eng.calls.count("integrate"), "force_local": eng.calls.count("force_local")}
    gathered = [None] * P
    dist.all_gather_object(gathered, res)
    if r == 0:
        json.dump(gathered, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
