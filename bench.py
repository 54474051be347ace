#!/usr/bin/env python3
"""bench.py — headline benchmark of the Barnes-Hut per-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver under torch.distributed.run, one rank per GPU over RCCL)

A "step" = one pass of the whole reference step (nbody_v5_bench.cu:255-283: bbox -> Morton keys
-> sort -> octree build -> COM -> force traversal -> integrate) over the synthetic workload
BASELINE.json's metric is quoted on: 1,000,000 bodies per GPU, Plummer sphere, theta = 0.5, fp32.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def b_alg(V, O, P, n):
    """24 B per cell MAC, 32 B per opened cell, 16 B per body interaction,
    24 B per body for its own load + store."""
    return 24 * V + 32 * O + 16 * P + 24 * n


def load_traffic(n, theta):
    """HBM bytes per force launch from committed PMC counters (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "force_traffic.json")
    try:
        t = json.load(open(path))
        for rec in t.get("records", []):
            if rec.get("n") == n and abs(rec.get("theta", -1) - theta) < 1e-6:
                return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def measured_copy_bandwidth(torch, nbytes=1 << 30, reps=10):
    """device-to-device copy of 1 GiB (read + write) timed with events: the practical HBM ceiling of this
    box next to the nominal 8 TB/s"""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.zero_()
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbs = 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    return gbs


def load_valu_insts(n, theta):
    """VALU instructions per force launch from the committed SQ counter pass (profiles/), or None."""
    if n != 1_000_000 or abs(theta - 0.5) > 1e-6:
        return None
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_v4", "force_fast_kernel_sq.json")))
        return float(t["counters"]["SQ_INSTS_VALU"]["mean"])
    except Exception:
        return None


def cpu_baseline(pkg, n, theta, ic, budget_s=25.0):
    """The CPU oracle ("port": this repo's restatement of the reference recurrence — the reference
    has no CPU path) timed on this box's host cores on the same workload: whole steps, all stages,
    OpenMP over all cores, until ~budget_s of wall time is used (at least one step)."""
    import oracle as O
    O.build()
    cores = O.max_threads()
    p = O.params(theta=theta)
    st = O.Oracle(n, p)
    st.upload(*ic)
    t0 = time.time()
    steps = 0
    per = []
    while True:
        t1 = time.time()
        st.step(1, order=O.ORDER_PREORDER)
        per.append(time.time() - t1)
        steps += 1
        if time.time() - t0 + per[-1] > budget_s or steps >= 5:
            break
    tm = st.times()
    cnt = st.counts()
    st.close()
    t_step = min(per)
    return {
        "value": n / t_step, "unit": "particles/s/step", "cores": cores, "kind": "port",
        "sample": f"{steps} whole step(s) of the same {n}-body theta={theta} workload, all stages, "
                  f"OpenMP x{cores}; best step {t_step:.3f} s (force {tm['force']:.3f} s)",
        "ms_per_step": t_step * 1e3,
        "oracle_counts_per_body": {"V": cnt["V"] / n, "O": cnt["O"] / n, "P": cnt["P"] / n},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bodies", "--n", dest="n", type=int, default=1_000_000,
                    help="bodies per GPU (weak scaling); use --bodies under torch.distributed.run")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --bodies is the TOTAL over all GPUs (default: weak, bodies per GPU)")
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--xcd-mode", type=int, default=0, help="tuning: block->chunk placement (bh_params.xcd_mode)")
    ap.add_argument("--leaf-cap", type=int, default=1, help="tuning: bodies per leaf (1 = reference intent)")
    ap.add_argument("--force-block", type=int, default=0, help="tuning: threads per force workgroup (64/128/256)")
    ap.add_argument("--force-variant", type=int, default=0,
                    help="A/B: 0 = hand-scheduled force walk (default), 1 = compiler-scheduled walk")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bhpkg
    pkg = bhpkg.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box (tests/test_gpu_dist.py): all ranks on GPU 0, gloo
    # standing in for RCCL.  Never set by the driver; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("BH_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dist = None
    # BH_FORCE_DIST=1 runs the multi-rank code path (RCCL init, sharded stepper, all-gather) even at
    # world size 1, so it can be exercised on a one-GPU box (tests/test_gpu_dist.py)
    multi = world > 1 or os.environ.get("BH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    n_total = args.n if args.strong else args.n * world  # default weak scaling: fixed bodies per GPU
    ic = pkg.plummer(n_total, seed=args.seed)  # identical on every rank (counter-based RNG)

    from nbody_barnes_hut_cuda_amd import dist as bhdist
    # N > 1: domain-decomposed stepping (each rank owns one Morton-key range, builds only its own
    # octree and imports locally-essential records).  BH_DIST_MODE=replicated selects the
    # round-1 scheme (replicated tree, sharded traversal, acc all-gather) for A/B.
    dist_mode = os.environ.get("BH_DIST_MODE", "domain") if multi else "single"
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def replicated():
        e, st = bhdist.make_gpu_stepper(pkg, n_total, device=local_rank, theta=args.theta,
                                        xcd_mode=args.xcd_mode,
                                        leaf_cap=args.leaf_cap, force_block=args.force_block,
                                        force_variant=args.force_variant)
        e.upload(*ic)
        return e, st

    fallback_reason = None
    if dist_mode == "domain":
        # A failure every rank hits alike (the size negotiations are functions of all-gathered data, so
        # capacity errors are) drops all ranks to the replicated scheme together instead of losing the run.
        try:
            stepper = bhdist.DomainStepper(pkg, ic, bhdist.TorchComm(), local_rank, theta=args.theta,
                                           xcd_mode=args.xcd_mode,
                                           force_block=args.force_block, force_variant=args.force_variant)
            eng = stepper.e
            stepper.step(args.warmup)
        except Exception as ex:  # noqa: BLE001
            fallback_reason = repr(ex)
            print(f"[bench rank {rank}] domain-decomposed stepping failed ({fallback_reason}); "
                  "falling back to the replicated scheme", file=sys.stderr, flush=True)
            dist_mode = "replicated"
            eng, stepper = replicated()
            stepper.step(args.warmup)
    else:
        eng, stepper = replicated()
        stepper.step(args.warmup)
    barrier()

    # algorithmic bytes of one force launch on the tree the timed region starts from
    counts0 = None
    if rank == 0 and not multi:
        eng.tree_stages()
        eng.force_count()
        s = eng.stats()
        counts0 = (s.count_V, s.count_O, s.count_P)
    eng.set_timing(not multi)  # per-step hipEvent pairs on the engine's stream (1-GPU path)

    if dist_mode == "domain" and rank == 0:
        stepper.set_profile(True)   # per-phase event pairs on rank 0's stream (a few microseconds per step)
    barrier()
    t0 = time.perf_counter()
    if not multi:
        eng.step(args.steps)   # bh_step: the C-ABI's own fused stage sequence
    else:
        try:
            stepper.step(args.steps)
        except Exception as ex:  # noqa: BLE001 - collective by construction (see DomainStepper.step)
            if dist_mode != "domain":
                raise
            fallback_reason = repr(ex)
            print(f"[bench rank {rank}] domain-decomposed stepping failed in the timed region "
                  f"({fallback_reason}); re-timing with the replicated scheme", file=sys.stderr, flush=True)
            dist_mode = "replicated"
            eng, stepper = replicated()
            stepper.step(args.warmup)
            barrier()
            t0 = time.perf_counter()
            stepper.step(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        st = eng.stats()
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_total * args.steps / elapsed
        roofline = None
        stages = None
        if not multi:
            f_ms, s_ms = eng.timing_history()
            eng.set_timing(False)
            eng.tree_stages()
            eng.force_count()
            s1 = eng.stats()
            counts1 = (s1.count_V, s1.count_O, s1.count_P)
            V, O, P = [(a + b) / 2.0 for a, b in zip(counts0, counts1)]
            bytes_alg = b_alg(V, O, P, n_total)
            avg_force_ms = float(np.mean(f_ms))
            achieved = bytes_alg / (avg_force_ms * 1e-3) / 1e9
            traffic = load_traffic(n_total, args.theta)
            valu = load_valu_insts(n_total, args.theta)
            roofline = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "force_fast_kernel",
                "avg_launch_ms": avg_force_ms,
                "launches_timed": int(len(f_ms)),
                "algorithmic_bytes_per_launch": bytes_alg,
                "per_body": {"V": V / n_total, "O": O / n_total, "P": P / n_total,
                             "bytes": bytes_alg / n_total},
                "hbm_copy_measured_GBs": measured_copy_bandwidth(torch),
                "valu_issue_frac": (valu * 4.0 / (1024 * avg_force_ms * 1e-3 * 2.4e9)) if valu else None,
                "limiter": "instruction issue: 26 instructions (16 VALU) per (record, wave), 57.8M such pairs on "
                           "1024 SIMDs vs a 21-29 ns memory-free microbenchmark floor of the 15-VALU body "
                           "(DESIGN.md §4, tools/ubench_valu.hip); valu_issue_frac = committed SQ_INSTS_VALU x 4 cycles / "
                           "(1024 SIMDs x this run's launch time x 2.4 GHz): the vector ALU is the saturated unit; "
                           "HBM is ~0.4 % utilised",
                "note": "algorithmic = per-lane no-reuse bytes of the recurrence (SURVEY 8d); the "
                        "wave-cooperative kernel fetches each record once per 64 lanes and the tree is "
                        "cache-resident, so this can exceed the HBM peak; 'traffic' is the measured HBM side",
            }
            stages = {"avg_force_ms": avg_force_ms, "avg_step_ms_device": float(np.mean(s_ms)),
                      "last_step_ms": {"bbox": st.ms_bbox, "morton": st.ms_morton, "sort": st.ms_sort,
                                       "build": st.ms_build, "com": st.ms_com, "force": st.ms_force,
                                       "integrate": st.ms_integrate}}
        out = {
            "metric": "particles/sec/step (1M bodies per GPU, theta=0.5)",
            "value": value, "unit": "particles/s/step",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{n_total // world:,} bodies per GPU ({n_total:,} total), Plummer sphere a=400 seed {args.seed}, "
                            f"theta={args.theta}, G=0.5 eps2=50 dt=0.02, fp32, leaf_cap=1, 63-bit keys "
                            "(BASELINE.json configs[2]; x8 = configs[3])",
                "n_total": n_total,
                "parallelism": "1 GPU" if not multi else (
                    f"{world} ranks, domain decomposition: per-rank octree of one Morton-key range, top tree + "
                    "locally-essential records by 4 all-gathers per step (RCCL)" if dist_mode == "domain" else
                    f"{world} ranks: replicated tree, Morton-slab sharded traversal, acc all-gather (RCCL)"),
                "tree": {"cells": st.n_internal, "records": st.n_entries, "max_level": st.max_level},
            },
            "roofline": roofline,
        }
        if stages:
            out["stages"] = stages
        if fallback_reason:
            out["config"]["domain_fallback"] = fallback_reason
        if dist_mode == "domain":
            out["config"]["domain"] = {
                "bodies_rank0": int(stepper.n_loc), "let_records_per_rank": [int(v) for v in stepper.let_counts],
                "let_stride": int(stepper.stride), "emigrants_last_step_max": int(stepper.mig_last),
                "let_retries": int(stepper.let_retries), "extra_migration_rounds": int(stepper.mig_rounds),
                "phase_ms_rank0": stepper.phase_ms()}
        assert st.status_flags == 0, st.status_flags
    if dist is not None:
        dist.barrier()
    if rank == 0 and not multi and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, n_total, args.theta, ic)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
