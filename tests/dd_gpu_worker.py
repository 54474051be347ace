"""Worker for tests/test_gpu_dist.py::test_domain_stepper_multiprocess_one_gpu — run under
torch.distributed.run.  Every rank drives dist.DomainStepper on cuda:0 (the ranks share the one GPU
of the test box) with the gloo backend standing in for RCCL; rank 0 gathers all local states and
compares them with a single-context bh_step run of the same initial conditions."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402


def main():
    out_path, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    split = {"one": False, "two": True, "auto": None, "adaptive": "adaptive"}[sys.argv[4] if len(sys.argv) > 4 else "auto"]
    split_pct = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    pkg = bhpkg.load()
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    world, rank = dist.get_world_size(), dist.get_rank()
    ic = pkg.plummer(n, seed=21)
    st = bhdist.DomainStepper(pkg, ic, bhdist.TorchComm(), 0, split=split, split_pct=split_pct)
    st.step(steps)
    ids, posm, vel, acc = st.local_state()
    flags = st.e.stats().status_flags
    gathered = [None] * world
    dist.all_gather_object(gathered, (ids, posm[:, :3].copy(), acc, int(flags)))
    if rank == 0:
        pos = np.zeros((n, 3), np.float32)
        a = np.zeros((n, 3), np.float32)
        seen = np.zeros(n, np.int64)
        fl = 0
        for i, p, ac, f in gathered:
            pos[i] = p
            a[i] = ac
            seen[i] += 1
            fl |= f
        with pkg.Engine(n) as e:
            e.upload(*ic)
            e.step(steps)
            x, y, z, *_ = e.download()
            ax, ay, az = e.download_acc()
        p1 = np.stack([x, y, z], 1)
        a1 = np.stack([ax, ay, az], 1)
        rel = np.linalg.norm(a - a1, axis=1) / np.maximum(np.linalg.norm(a1, axis=1), 1e-30)
        st._sync_info()
        json.dump({"world": world, "owned_once": bool((seen == 1).all()), "flags": fl,
                   "split_now": st.split_now, "x4_us": st.x4_us,
                   "max_dpos": float(np.abs(pos - p1).max()), "acc_rel_median": float(np.median(rel)),
                   "acc_rel_max": float(rel.max())}, open(out_path, "w"))
    dist.barrier()
    st.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
