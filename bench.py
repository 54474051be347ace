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

# RCCL between processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails otherwise); harmless at N = 1
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VECTOR_TFLOPS = 157.3  # MI355X peak FP32 vector rate (same guide): 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz

# Issue cost of the force walk's instruction FORMS: cycles one SIMD needs per wave64 instruction with 8 waves
# resident, measured on MI355X by tools/ubench_forms.hip / tools/ubench_issue.hip (loop overhead removed);
# committed output: profiles/r02_ubench/ubench_forms.txt, ubench_issue.txt.  Only two-source all-VGPR VOP2 runs
# at the 2.3-cycle full rate; SGPR operands, three sources, SGPR results and packed fp32 cost 3.8-4.3.
# "scalar_in_context": what one more SALU instruction per pair chain costs the SIMD while seven other waves issue
# VALU work (tools/ubench_exec.hip, profiles/r03_ubench/ubench_exec.txt: pair chain + 0 / 2 / 4 / 8 SALU =
# 73.8 / 79.0 / 81.0 / 83.7 cycles): scalar issue overlaps other waves' vector issue to a large part, so it is
# NOT added to the floor (round 2 added 2.2 cycles per scalar instruction and called the sum a floor; the 8M and
# theta = 0.3 configurations ran faster than that "floor").
ISSUE_CYCLES = {"pk_add_sgpr": 3.78, "pk_mul": 3.78, "pk_fma": 4.30, "cmp_e64": 4.30, "cndmask_e64": 4.30,
                "rsq": 7.76, "lane_rw": 2.80, "scalar_in_context": 1.25}
ISSUE_PROVENANCE = ("committed profiles r02_ubench (tools/ubench_forms.hip, tools/ubench_issue.hip) and r03_ubench "
                    "(tools/ubench_exec.hip) on MI355X")


def b_alg(V, O, P, n):
    """24 B per cell MAC, 32 B per opened cell, 16 B per body interaction,
    24 B per body for its own load + store."""
    return 24 * V + 32 * O + 16 * P + 24 * n


def csrc_sha16():
    """identity of the kernels this run executes: sha256 over the library's sources (the GPU box has no .git)"""
    import hashlib
    d = os.path.join(ROOT, "nbody-barnes-hut-cuda_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(n, theta):
    """(HBM bytes per force launch, provenance, stale) from the committed PMC passes under profiles/, or (None, None,
    None).  Not measured by this run: rocprofv3 counters cannot be read from inside the benchmark.  `stale` = the
    record was taken with other kernel sources than this run's (tools/collect_profiles.py stores the csrc hash the
    profiled bench printed)."""
    path = os.path.join(ROOT, "profiles", "force_traffic.json")
    try:
        t = json.load(open(path))
        for rec in reversed(t.get("records", [])):
            if rec.get("n") == n and abs(rec.get("theta", -1) - theta) < 1e-6:
                return (rec.get("hbm_bytes_per_launch"),
                        f"committed profile {rec.get('profile')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)",
                        rec.get("csrc_sha16") != csrc_sha16())
    except Exception:
        pass
    return None, None, None


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


def issue_roofline(ws, simds, launch_ms):
    """(The counters are those of the one-wave-per-group walk over ALL groups; the groups a launch walks
    cooperatively evaluate the same record pairs and differ in how a stack entry travels — LDS instead of lanes.)
    VALU-issue floor of one force launch from the walk's own event counters (bh_force_walk_stats, counted in
    THIS run) priced with the per-form issue costs above.  Instruction counts follow csrc/bh_force.hip:
    a pair = 3 v_pk_add + 3 v_pk_fma + 2 v_cmp + 2 v_rsq + 3 v_pk_mul + 3 v_pk_fma (+ 2 v_cndmask if a record was
    opened) and 2 scalar (s_or + branch); a block = 2-4 s_load + ~13 scalar / branch; a push = ~9 scalar; a stack
    entry that is not its block's last push goes through the lanes: 3 v_writelane + 3 v_readlane (counted:
    lane_spills; the last push of a block stays in scalar registers).  The BOUND is the VALU-only figure (no configuration can beat it); the estimate
    with scalar issue at its measured in-context cost is reported beside it and is an estimate, not a floor."""
    c = ISSUE_CYCLES
    pairs, blocks, masked, waves = float(ws.pairs), float(ws.blocks), float(ws.masked_pairs), float(ws.waves)
    pushes = blocks - waves
    pair_cycles = 3 * c["pk_add_sgpr"] + 6 * c["pk_fma"] + 2 * c["cmp_e64"] + 2 * c["rsq"] + 3 * c["pk_mul"]
    spills = float(getattr(ws, "lane_spills", 0))
    skipped = float(getattr(ws, "no_taker_pairs", 0))  # pairs whose force half (3 v_pk_mul + 3 v_pk_fma) is not issued
    valu_insts = 16 * pairs + 2 * masked + 6 * spills - 6 * skipped
    valu_cycles = (pair_cycles * pairs + 2 * c["cndmask_e64"] * masked + c["lane_rw"] * 6 * spills
                   - (3 * c["pk_mul"] + 3 * c["pk_fma"]) * skipped)
    scalar_insts = 2 * pairs + 13 * blocks + 9 * pushes
    clock_hz = ws.clock_ghz * 1e9
    floor_valu_ms = valu_cycles / simds / clock_hz * 1e3
    est_ms = (valu_cycles + c["scalar_in_context"] * scalar_insts) / simds / clock_hz * 1e3
    return {
        "counted_this_run": {"waves": int(waves), "record_pairs": int(pairs), "blocks": int(blocks),
                             "pairs_with_opened_record": int(masked), "stack_entries_through_lanes": int(spills),
                             "pairs_nobody_takes_force_half_skipped": int(skipped)},
        "valu_insts_per_launch": valu_insts, "scalar_insts_per_launch": scalar_insts,
        "cycles_per_pair_valu": pair_cycles, "issue_cycles_per_form": c, "issue_cycles_provenance": ISSUE_PROVENANCE,
        "clock_ghz_in_kernel": ws.clock_ghz, "simds": simds,
        "floor_ms_valu_only": floor_valu_ms, "kernel_ms": launch_ms, "frac_of_valu_floor": floor_valu_ms / launch_ms,
        "estimate_ms_valu_plus_scalar_in_context": est_ms,
        "estimate_note": "VALU floor + scalar instructions at their measured marginal cost beside other waves' VALU "
                         "issue; an estimate of the issue time, not a bound (launch ramp / drain, scalar-load waits "
                         "and lone-wave issue rates at the end of the launch are not in it)",
        "wave_lifetime_ms": {"mean": ws.wave_cycles_mean / clock_hz * 1e3, "max": ws.wave_cycles_max / clock_hz * 1e3},
    }


def residency_from_trace(rows, slots):
    """rows: uint32[waves, 4] of Engine.force_launch_trace (start, end on the 100 MHz clock, HW_ID, XCC_ID).
    -> what the launch did with the GPU's wave slots: Σ wave lifetime / span = mean resident waves,
    when the last wave started, how many waves were resident along the launch, how long the SIMDs idle at its end."""
    import numpy as np
    if rows is None or len(rows) == 0:
        return None
    t0 = rows[:, 0].astype(np.int64)
    t1 = rows[:, 1].astype(np.int64)
    base = t0.min()
    t0 = (t0 - base) * 0.01  # us
    t1 = (t1 - base) * 0.01
    span = float(t1.max())
    hw, xcc = rows[:, 2], rows[:, 3] & 0xF
    key = ((xcc.astype(np.int64) * 16 + ((hw >> 12) & 0xF)) * 16 + ((hw >> 8) & 0xF)) * 4 + ((hw >> 4) & 3)
    uk, inv = np.unique(key, return_inverse=True)
    last_end = np.zeros(len(uk))
    np.maximum.at(last_end, inv, t1)
    W = len(t0)
    ev = np.concatenate([np.stack([t0, np.ones(W)], 1), np.stack([t1, -np.ones(W)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    res = np.cumsum(ev[:, 1])
    tt = ev[:, 0]
    at = {}
    for frac in (0.25, 0.5, 0.75, 0.9, 0.95):
        k = min(int(np.searchsorted(tt, frac * span)), len(res) - 1)
        at[f"{frac:.2f}"] = int(res[k])
    life = t1 - t0
    mean_res = float(life.sum() / span)
    return {"waves": int(W), "simds_seen": int(len(uk)), "wave_slots": int(slots), "span_us": span,
            "mean_resident_waves": mean_res, "mean_resident_frac_of_slots": mean_res / slots,
            "peak_resident_waves": int(res.max()),
            "last_wave_start_frac_of_span": float(t0.max() / span),
            "resident_waves_at_frac_of_span": at,
            "wave_lifetime_us": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)),
                                 "p90": float(np.percentile(life, 90)), "max": float(life.max())},
            "simd_idle_before_launch_end_us": {"mean": float((span - last_end).mean()),
                                               "max": float((span - last_end).max())},
            "note": "one traced launch of the same kernel, grid and placement as the timed steps' force launch, not "
                    "fused with the integrate step (bh_force_launch_trace)"}


def _oracle_steps(O, n, theta, ic, nthreads, budget_s, max_steps):
    p = O.params(theta=theta)
    st = O.Oracle(n, p)
    st.upload(*ic)
    t0 = time.time()
    per = []
    while True:
        t1 = time.time()
        st.step(1, order=O.ORDER_PREORDER, nthreads=nthreads)
        per.append(time.time() - t1)
        if time.time() - t0 + per[-1] > budget_s or len(per) >= max_steps:
            break
    tm = st.times()
    cnt = st.counts()
    st.close()
    return per, tm, cnt


def cpu_baseline_child(n, theta, seed, budget_s, max_steps):
    """One leg of the CPU baseline, run in a CHILD process of bench.py so that the OpenMP runtime starts with
    this leg's own team size and binding (OMP_NUM_THREADS / OMP_PROC_BIND=close / OMP_PLACES=cores are set by the
    parent; an OpenMP runtime that torch already initialised in the parent would ignore them).  Never touches the GPU."""
    import statistics
    import bhpkg
    import oracle as O
    pkg = bhpkg.load()
    O.build()
    ic = pkg.plummer(n, seed=seed)
    per, tm, cnt = _oracle_steps(O, n, theta, ic, 0, budget_s, max_steps)
    print(json.dumps({"threads": O.max_threads(), "best_s": min(per), "median_s": statistics.median(per), "steps": len(per),
                      "force_s_last": tm["force"], "counts": cnt}))


def _cpu_leg(n, theta, seed, threads, budget_s, max_steps):
    import subprocess
    env = dict(os.environ)
    env.update({"OMP_NUM_THREADS": str(threads), "OMP_PROC_BIND": "close", "OMP_PLACES": "cores"})
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--bodies", str(n),
                          "--theta", str(theta), "--seed", str(seed), "--child-budget", str(budget_s),
                          "--child-steps", str(max_steps)], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline leg failed: " + out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_baseline(n, theta, seed):
    """The CPU oracle ("port": this repo's restatement of the reference recurrence — the reference has no CPU
    path; gcc -O3, iterative walk, OpenMP) timed on this box's host cores: whole steps, all stages.  `value` =
    the bench workload on all cores; plus BASELINE.json configs[0] (65,536 bodies) on a team sized to the problem
    (one thread per 2,048 bodies, at most all cores: 128 threads on a 65k-body problem measured 6x apart between
    two boxes in round 2) and on ONE thread.  Threads are bound (OMP_PROC_BIND=close, OMP_PLACES=cores);
    best and median step are both reported.  Bounded: about 25 s of CPU wall time in total."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    r = _cpu_leg(n, theta, seed, cores, 16.0, 5)
    out = {
        "value": n / r["best_s"], "unit": "particles/s/step", "cores": r["threads"], "kind": "port",
        "sample": f"{r['steps']} whole step(s) of the same {n}-body theta={theta} workload, all stages, "
                  f"OpenMP x{r['threads']} bound to cores; best step {r['best_s']:.3f} s, median {r['median_s']:.3f} s "
                  f"(force stage of the last step {r['force_s_last']:.3f} s); the oracle is built with gcc -O3 "
                  "-fopenmp -ffp-contract=off and NOT -march=native (BASELINE.md §2.1 names it: the .so is built "
                  "in the build container and travels to the GPU box, so it targets baseline x86-64)",
        "ms_per_step": r["best_s"] * 1e3, "ms_per_step_median": r["median_s"] * 1e3,
        "oracle_counts_per_body": {k: r["counts"][k] / n for k in ("V", "O", "P")},
    }
    n1 = 65536
    team = max(1, min(cores, n1 // 2048))
    ra = _cpu_leg(n1, 0.5, 42, team, 3.0, 20)
    r1 = _cpu_leg(n1, 0.5, 42, 1, 6.0, 3)
    out["config0_65536_bodies_theta0.5"] = {
        "team": {"value": n1 / ra["median_s"], "unit": "particles/s/step", "cores": ra["threads"],
                 "ms_per_step": ra["median_s"] * 1e3, "ms_per_step_best": ra["best_s"] * 1e3,
                 "sample": f"median of {ra['steps']} whole steps, {ra['threads']} threads = one per 2,048 bodies, bound to "
                           f"cores (force stage of the last step {ra['force_s_last'] * 1e3:.1f} ms)"},
        "one_thread": {"value": n1 / r1["median_s"], "unit": "particles/s/step", "cores": 1,
                       "ms_per_step": r1["median_s"] * 1e3, "ms_per_step_best": r1["best_s"] * 1e3,
                       "sample": f"median of {r1['steps']} whole steps (force stage of the last step {r1['force_s_last'] * 1e3:.1f} ms)"},
    }
    return out


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
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-budget", type=float, default=10.0, help=argparse.SUPPRESS)
    ap.add_argument("--child-steps", type=int, default=5, help=argparse.SUPPRESS)
    ap.add_argument("--xcd-mode", type=int, default=3, help="tuning: block->chunk placement (bh_params.xcd_mode, 3 = automatic)")
    ap.add_argument("--leaf-cap", type=int, default=1, help="tuning: bodies per leaf (1 = reference intent)")
    ap.add_argument("--force-block", type=int, default=0, help="tuning: threads per force workgroup (64/128/256)")
    ap.add_argument("--force-coop", type=int, default=0,
                    help="tuning: waves per group of the force walk (bh_params.force_coop; 0 = automatic, 1 = one wave per group)")
    ap.add_argument("--dd-split", action="store_true",
                    help="N > 1: the first 30 %% of a rank's bodies in two force passes (own pieces while X4 travels, "
                         "then the remote pass) whatever X4 costs; default: adaptive (one pass while the measured X4 "
                         "is short)")
    ap.add_argument("--dd-one-pass", action="store_true", help="N > 1: one force pass per step, after X4, always")
    ap.add_argument("--graph", action="store_true",
                    help="time bh_step replayed as a HIP graph (no per-stage event records inside the timed region; "
                         "the force-launch time of the roofline block then comes from 10 extra timed-stage steps)")
    ap.add_argument("--force-variant", type=int, default=0,
                    help="A/B: 0 = hand-scheduled force walk (default), 1 = compiler-scheduled walk")
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.n, args.theta, args.seed, args.child_budget, args.child_steps)
    # stdout carries exactly ONE line, the JSON: libraries that print banners to fd 1 (RCCL prints its version
    # block at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
        # a finite collective timeout: a rank stuck in (or missing from) an exchange ends every rank with a
        # non-zero exit (torch's watchdog aborts the process) instead of hanging the node
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("BH_COLLECTIVE_TIMEOUT_S", "180")))
        if rehearsal:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=tmo)

    n_total = args.n if args.strong else args.n * world  # default weak scaling: fixed bodies per GPU
    ic = pkg.plummer(n_total, seed=args.seed)  # identical on every rank (counter-based RNG)

    from nbody_barnes_hut_cuda_amd import dist as bhdist
    # N > 1: domain-decomposed stepping (each rank owns one interval of the key curve, builds only its own
    # octree and imports locally-essential records).  BH_DIST_MODE=replicated selects the
    # round-1 scheme (replicated tree, sharded traversal, acc all-gather) for A/B.
    dist_mode = os.environ.get("BH_DIST_MODE", "domain") if multi else "single"
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if dist is None:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    engine_kw = dict(theta=args.theta, xcd_mode=args.xcd_mode, force_block=args.force_block,
                     force_variant=args.force_variant)

    # The 1-GPU reference of the scaling figures, measured in THIS run before anything collective: bh_step on
    # --bodies bodies on every GPU at the same time (the node's power / clock state of the multi-GPU job), the
    # slowest rank counts.  aggregate_x = value / (bodies / that time).
    n1 = None
    if multi:
        icn = pkg.plummer(args.n, seed=args.seed)
        with pkg.Engine(args.n, device=local_rank, leaf_cap=args.leaf_cap, force_coop=args.force_coop, **engine_kw) as e1:
            e1.upload(*icn)
            e1.step(args.warmup)
            e1.sync()
            barrier()
            t1 = time.perf_counter()
            e1.step(args.steps)
            e1.sync()
            mine = (time.perf_counter() - t1) * 1e3 / args.steps
        n1 = {"bodies": args.n, "ms_per_step_slowest_gpu": reduce_max(mine), "ms_per_step_rank0": mine,
              "note": "bh_step on one GPU's share, all GPUs of the job running it at once, same process, before "
                      "the collective part"}
        del icn

    # how the exchanges travel: the library's RCCL transport on the rank's own stream (bh_comm_rccl_init_rank, unique
    # id broadcast through torch.distributed); every rank must get it or all use the torch.distributed callbacks
    transport = None
    def make_comm():
        nonlocal transport
        if rehearsal or os.environ.get("BH_BENCH_COMM") == "torch":
            transport = "torch.distributed callbacks (%s)" % dist.get_backend()
            return bhdist.TorchComm()
        comm, ok = None, 1.0
        try:
            comm = bhdist.RcclComm(local_rank)
            if not comm.check():   # collective; a transport that misdelivers is as good as none
                raise RuntimeError("bh_comm_check failed")
        except Exception as ex:  # noqa: BLE001 - decided collectively below
            print(f"[bench rank {rank}] RCCL transport unavailable: {ex!r}", file=sys.stderr, flush=True)
            ok = 0.0
        if reduce_max(1.0 - ok) > 0.0:   # somebody failed: everybody falls back
            transport = "torch.distributed callbacks (nccl); the library's RCCL transport failed to initialise"
            return bhdist.TorchComm()
        transport = "library RCCL transport (ncclAllGather / ncclAllToAll on the rank's stream)"
        return comm

    def domain_stepper(ic_, let_cap=None, slack=1.3):
        return bhdist.DomainStepper(pkg, ic_, make_comm(), local_rank, let_cap=let_cap, slack=slack,
                                    split=True if args.dd_split else (False if args.dd_one_pass else "adaptive"), **engine_kw)

    def replicated():
        e, st = bhdist.make_gpu_stepper(pkg, n_total, device=local_rank, leaf_cap=args.leaf_cap,
                                        force_coop=args.force_coop, step_graph=1 if args.graph else 0, **engine_kw)
        e.upload(*ic)
        return e, st

    fallback_reason = None
    if dist_mode == "domain":
        # A failure every rank hits alike (the size negotiations are functions of all-gathered data, so
        # capacity errors are) drops all ranks to the replicated scheme together instead of losing the run.
        try:
            # BH_BENCH_LET_CAP: rehearsal hook (tests/test_gpu_dist.py drives the fallback with a tiny LET capacity)
            let_cap = int(os.environ["BH_BENCH_LET_CAP"]) if rehearsal and "BH_BENCH_LET_CAP" in os.environ else None
            try:
                stepper = domain_stepper(ic, let_cap)
                eng = stepper.e
                stepper.step(args.warmup)
            except bhdist.DomainLeft as ex:
                if let_cap is not None:   # (the rehearsal of the fall-back wants it)
                    raise
                # every rank is here together: once more with room for 70 % more bodies and records per rank (a
                # rank's share can outgrow the 30 % slack on strongly clustered input) before giving up the scheme
                print(f"[bench rank {rank}] {ex!r}; retrying with slack 1.7", file=sys.stderr, flush=True)
                stepper.close()
                stepper = domain_stepper(ic, None, slack=1.7)
                eng = stepper.e
                stepper.step(args.warmup)
        except bhdist.DomainLeft as ex:  # raised on every rank after the same exchange; anything else is rank-local
            fallback_reason = repr(ex)   # and must fail the run (a mismatched collective would hang the others)
            print(f"[bench rank {rank}] domain-decomposed stepping failed ({fallback_reason}); "
                  "falling back to the replicated scheme", file=sys.stderr, flush=True)
            dist_mode = "replicated"
            eng, stepper = replicated()
            stepper.step(args.warmup)
    elif multi:
        eng, stepper = replicated()
        stepper.step(args.warmup)
    else:
        eng = pkg.Engine(n_total, device=local_rank, leaf_cap=args.leaf_cap, force_coop=args.force_coop,
                         step_graph=1 if args.graph else 0, **engine_kw)
        eng.upload(*ic)
        eng.step(args.warmup)
    barrier()

    # algorithmic bytes of one force launch on the tree the timed region starts from
    counts0 = None
    if rank == 0 and not multi:
        eng.tree_stages()
        eng.force_count()
        s = eng.stats()
        counts0 = (s.count_V, s.count_O, s.count_P)
    # 1-GPU path: a hipEvent pair around the force launch of every 4th step, on the engine's stream, inside the
    # timed region (each record costs the stream ~7 us: a pair per step is 1 % of a 1M-body step and 4 % of a
    # 65,536-body one; the per-stage breakdown comes from 10 further steps with an event after every stage)
    eng.set_timing(3 if (not multi and not args.graph) else 0)

    barrier()
    t0 = time.perf_counter()
    if not multi:
        eng.step(args.steps)   # bh_step: the C-ABI's own fused stage sequence
    else:
        try:
            stepper.step(args.steps)
        except bhdist.DomainLeft as ex:  # collective by construction (see DomainStepper.step)
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
    elapsed = reduce_max(time.perf_counter() - t0)

    # per-phase device times of the domain-decomposed step: 5 further steps on every rank (the exchanges are
    # collective), events on rank 0's stream only — NOT in the timed region: each event record costs the stream
    # several microseconds and the slowest rank sets the job's time
    if dist_mode == "domain":
        stepper.set_profile(rank == 0)
        try:
            stepper.step(5)
        except bhdist.DomainLeft as ex:
            print(f"[bench rank {rank}] profiling steps left the domain scheme: {ex!r}", file=sys.stderr, flush=True)
        barrier()

    # BASELINE.json's metric is worded on strong scaling ("1M bodies ... at 1/2/4/8 GPUs"): the same --bodies as ONE
    # system over all GPUs, in the same invocation (the headline `value` stays the weak-scaling workload, configs[3])
    strong = None
    if multi and not args.strong and dist_mode == "domain" and os.environ.get("BH_BENCH_NO_STRONG") != "1":
        ic_s = pkg.plummer(args.n, seed=args.seed)
        st_s = None
        try:
            st_s = domain_stepper(ic_s)
            st_s.step(args.warmup)
            barrier()
            ts = time.perf_counter()
            st_s.step(args.steps)
            barrier()
            el = reduce_max(time.perf_counter() - ts)
            strong = {"n_total": args.n, "value": args.n * args.steps / el, "unit": "particles/s/step",
                      "ms_per_step": el * 1e3 / args.steps, "scaling": "strong",
                      "let_records_per_rank": [int(v) for v in st_s.let_counts], "bodies_rank0": int(st_s.n_loc)}
            if n1:
                strong["speedup_vs_one_gpu"] = n1["ms_per_step_slowest_gpu"] / strong["ms_per_step"]
        except bhdist.DomainLeft as ex:
            strong = {"n_total": args.n, "error": repr(ex)}
        finally:
            if st_s is not None:
                st_s.close()
        barrier()

    out = None
    if rank == 0:
        st = eng.stats()
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_total * args.steps / elapsed
        roofline = None
        stages = None
        if not multi:
            f_ms = None
            if not args.graph:
                f_ms, _ = eng.timing_history()   # the force launches of the timed region
            eng.set_timing(True)                 # stage times: 10 further steps with an event after every stage
            eng.step(10)
            eng.sync()
            f10, s_ms = eng.timing_history()
            st = eng.stats()
            if f_ms is None:   # the timed region ran as graph replays (no events inside a replayed graph)
                f_ms = f10
            eng.set_timing(False)
            eng.tree_stages()
            eng.force_count()
            s1 = eng.stats()
            counts1 = (s1.count_V, s1.count_O, s1.count_P)
            V, O, P = [(a + b) / 2.0 for a, b in zip(counts0, counts1)]
            bytes_alg = b_alg(V, O, P, n_total)
            avg_force_ms = float(np.mean(f_ms))
            ws = eng.force_walk_stats()          # the walk's event counters + in-kernel clock, counted now
            props = torch.cuda.get_device_properties(local_rank)
            issue = issue_roofline(ws, 4 * props.multi_processor_count, avg_force_ms)
            # what the launch does with the GPU's wave slots (8 per SIMD: the walk kernels use 80 scalar registers)
            issue["residency"] = residency_from_trace(eng.force_launch_trace(), 4 * props.multi_processor_count * 8)
            groups = (n_total + 63) // 64
            tail = 4 * props.multi_processor_count * 7 // 3
            if groups > 2 * tail:
                launch = (f"force_mixed_kernel<FUSE> (csrc/bh_force.hip): groups 0..{(groups - tail) // 4 * 4 - 1} one wave "
                          f"each (hand-scheduled depth-first walk), the last {groups - (groups - tail) // 4 * 4} groups four "
                          "waves each (cooperative level-by-level walk) so that the short jobs fill the slots the long "
                          "ones leave")
            else:
                launch = (f"force_coop_kernel<FUSE> (csrc/bh_force.hip): {groups} groups, "
                          f"{8 if groups * 8 <= 4 * props.multi_processor_count * 8 else 4} waves per group "
                          "(cooperative level-by-level walk)")
            # useful arithmetic of the recurrence (ref:205-213): 20 flop per interaction TAKEN by a body (accepted
            # cells V - O and body interactions P: 3 sub, 5 for d2 + eps2, rsq, 2 for the MAC, 3 for f, 6 for the
            # accumulate) and 11 per cell a body OPENS (the MAC half only: its force half is discarded)
            useful_flop = (V - O + P) * 20.0 + O * 11.0
            useful_tflops = useful_flop / (avg_force_ms * 1e-3) / 1e12
            traffic, traffic_src, traffic_stale = load_traffic(n_total, args.theta)
            alg_gbs = bytes_alg / (avg_force_ms * 1e-3) / 1e9
            roofline = {
                "bound": "valu", "achieved": useful_tflops, "peak": FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": useful_tflops / FP32_VECTOR_TFLOPS,
                "traffic": traffic, "traffic_provenance": traffic_src, "traffic_stale": traffic_stale,
                "kernel": launch + "; in bh_step the launch also integrates its bodies and folds the next cube "
                          "(~3 us of the launch time)",
                "avg_launch_ms": avg_force_ms,
                "launches_timed": int(len(f_ms)),
                "why_valu": "the walk is bound by VALU instruction ISSUE: every record field arrives in SGPRs, and "
                            "SGPR-operand / packed / compare / rsq forms issue at 3.8-7.8 cycles per wave64 "
                            "instruction (full rate 2.3 only for 2-source all-VGPR ops); the 97 %-L2-resident tree "
                            "keeps HBM at a few per cent (DESIGN.md §4); frac = useful flop of the recurrence (20 per interaction "
                            "taken, 11 per cell opened) / FP32 vector peak, lane_efficiency = share of the issued lane "
                            "slots that carry a body's interaction",
                "issue": issue,
                "useful_flop_per_launch": useful_flop,
                "lane_efficiency": (V + P) / max(1.0, 2.0 * float(ws.pairs) * 64.0),
                "per_body": {"V": V / n_total, "O": O / n_total, "P": P / n_total, "bytes": bytes_alg / n_total},
                "hbm": {
                    "algorithmic_bytes_per_launch": bytes_alg, "algorithmic_GBs": alg_gbs,
                    "algorithmic_over_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                    "measured_traffic_GBs": (traffic / (avg_force_ms * 1e-3) / 1e9) if traffic else None,
                    "measured_traffic_over_hbm_peak": (traffic / (avg_force_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "hbm_copy_measured_GBs": measured_copy_bandwidth(torch),
                    "note": "informational: SURVEY §8(d)'s per-lane no-reuse byte model is not HBM traffic for a "
                            "wave-cooperative walk (one scalar fetch per record per 64 bodies, tree resident in "
                            "L2 / Infinity Cache), so it can exceed the HBM peak; the measured side is `traffic`",
                },
            }
            stages = {"avg_force_ms": avg_force_ms, "avg_step_ms_device_with_stage_events": float(np.mean(s_ms)),
                      "last_step_ms": {"bbox": st.ms_bbox, "morton": st.ms_morton, "sort": st.ms_sort,
                                       "build": st.ms_build, "com": st.ms_com, "force": st.ms_force,
                                       "integrate": st.ms_integrate}}
        per_gpu = n_total // world
        out = {
            "metric": (f"particles/sec/step ({n_total:,} bodies in total over {world} GPU(s), theta={args.theta})" if args.strong
                       else f"particles/sec/step ({per_gpu:,} bodies per GPU, theta={args.theta})"),
            "value": value, "unit": "particles/s/step",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{n_total // world:,} bodies per GPU ({n_total:,} total), Plummer sphere a=400 seed {args.seed}, "
                            f"theta={args.theta}, G=0.5 eps2=50 dt=0.02, fp32, leaf_cap=1, 63-bit keys in Hilbert order "
                            "(BASELINE.json configs[2]; x8 = configs[3])",
                "n_total": n_total,
                "parallelism": "1 GPU" if not multi else (
                    f"{world} ranks, domain decomposition: per-rank octree of one interval of the key curve, top tree + "
                    "locally-essential records: 3 all-gathers + 1 all-to-all per step, protocol inside libbh.so "
                    f"(bh_rank_step); transport: {transport}" if dist_mode == "domain" else
                    f"{world} ranks: replicated tree, Morton-slab sharded traversal, acc all-gather (RCCL)"),
                "tree": {"cells": st.n_internal, "records": st.n_entries, "max_level": st.max_level},
            },
            "roofline": roofline,
            "build": {"csrc_sha16": csrc_sha16(), "abi": int(pkg.lib.bh_abi_version())},
        }
        if n1:
            out["n1_ms_per_step"] = n1["ms_per_step_slowest_gpu"]
            out["n1"] = n1
            # whole-job throughput over the throughput of ONE GPU stepping one GPU's share (weak scaling: that share
            # is --bodies; --strong: the whole system on one GPU)
            out["aggregate_x"] = value / (n1["bodies"] / (n1["ms_per_step_slowest_gpu"] * 1e-3))
        if strong:
            out["strong"] = strong
        if stages:
            out["stages"] = stages
        if fallback_reason:
            out["config"]["domain_fallback"] = fallback_reason
        if dist_mode == "domain":
            out["config"]["domain"] = {
                "bodies_rank0": int(stepper.n_loc), "let_records_per_rank": [int(v) for v in stepper.let_counts],
                "let_stride": int(stepper.stride), "emigrants_last_step_max": int(stepper.mig_last),
                "x4": "per-destination segments, all-to-all" if stepper.let_mode == 1 else "union segment, all-gather",
                "force_passes": {0: "one, after X4", 1: "first part of the bodies in two (own pieces beside X4, then the "
                                 "remote pass), the rest in one", 2: "adaptive: one pass while the measured X4 is short, "
                                 "the split form from ~0.2 ms on"}[int(stepper.split)],
                "split_now_rank0": int(stepper.split_now),
                # adaptive form: the exchange as this rank's stream saw it (events around X4, running mean), or null
                "x4_ms_measured_rank0": (stepper.x4_us / 1000.0) if stepper.x4_us >= 0 else None,
                # per pair, what the pair needed in the last step + a quarter + 4096 records (bh_comm.all_to_all_v);
                # one size for all pairs (the slot) would be x4_bytes_slots
                "x4_bytes_received_per_gpu_per_step": int(stepper.x4_recv_bytes),
                "x4_bytes_slots": int(world * stepper.stride * 32),
                "let_retries": int(stepper.let_retries), "extra_migration_rounds": int(stepper.mig_rounds),
                "phase_ms_rank0": stepper.phase_ms()}
        assert st.status_flags == 0, st.status_flags
    if dist is not None:
        dist.barrier()
    if rank == 0 and not multi and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n_total, args.theta, args.seed)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_mode == "domain":
        stepper.close()   # the rank owns its context (and its RCCL communicator)
    else:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
