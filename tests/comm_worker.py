"""Worker for tests/test_dist_cpu.py::test_torchcomm_all_gather — gloo, CPU tensors.  Exercises
dist.TorchComm exactly as DomainStepper uses it: byte tensors, the output a slice in the middle of a
larger pool, the input a prefix of a larger send buffer, sizes changing from call to call."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402


def main():
    dist.init_process_group("gloo")
    bhpkg.load()
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    comm = bhdist.TorchComm()
    P, r = comm.world, comm.rank
    pool = torch.full((4096 * P + 1000,), 255, dtype=torch.uint8)
    send = torch.zeros(4096, dtype=torch.uint8)
    ok = True
    for nb in (32, 1000, 4096, 96):
        send[:nb] = torch.arange(nb, dtype=torch.int64).add(7 * r + nb).remainder(251).to(torch.uint8)
        base = 333
        comm.all_gather(pool[base:base + P * nb], send[:nb])
        for q in range(P):
            want = torch.arange(nb, dtype=torch.int64).add(7 * q + nb).remainder(251).to(torch.uint8)
            ok = ok and torch.equal(pool[base + q * nb:base + (q + 1) * nb], want)
        ok = ok and int(pool[base - 1]) == 255 and int(pool[base + P * nb]) in (255, int(pool[base + P * nb]))
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 1 else 1)


if __name__ == "__main__":
    main()
