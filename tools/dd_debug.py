"""Lockstep rehearsal of the domain-decomposed step on ONE GPU with per-step diagnostics:
python tools/dd_debug.py --world 4 --n 32768 --steps 10 [--stream-ic] [--check]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--stream-ic", action="store_true", help="two counter-streaming halves (heavy migration)")
    ap.add_argument("--mig-frac", type=float, default=0.5)
    ap.add_argument("--check", action="store_true", help="compare with a single context")
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--ic", default="plummer", choices=["plummer", "disc"])
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="one force pass after X4 instead of own + remote passes")
    ap.add_argument("--let-mode", type=int, default=None, help="X4: 0 all-gather of the union, 1 per-destination all-to-all (default: the stepper's)")
    ap.add_argument("--force-coop", type=int, default=0, help="bh_params.force_coop (1 = one wave per group)")
    args = ap.parse_args()
    import torch
    pkg = bhpkg.load()
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    n, P = args.n, args.world
    ic = [a.copy() for a in (pkg.plummer if args.ic == "plummer" else pkg.disc)(n, seed=args.seed)]
    if args.stream_ic:
        ic[3] += 400.0
        ic[3][: n // 2] -= 800.0
    ic = tuple(ic)
    order = bhdist.global_morton_order(pkg, ic, 0)
    group = bhdist.LocalGroup(P)
    stream = torch.cuda.Stream(0)
    steppers = [None] * P
    slow = [0] * P
    info = [None] * P
    errs = []

    def log(*a):
        print(*a, file=sys.stderr, flush=True)

    def work(r):
        try:
            torch.cuda.set_device(0)
            st = bhdist.DomainStepper(pkg, ic, bhdist.LocalComm(group, r), 0, stream=stream, order=order,
                                      mig_frac=args.mig_frac, split=not args.no_split, theta=args.theta,
                                      let_mode=args.let_mode, force_coop=args.force_coop)
            steppers[r] = st
            group.barrier.wait()
            for s in range(args.steps):
                t0 = time.time()
                st.step(1)
                sst = st.e.stats()
                flags = sst.status_flags
                slow[r] = sst.sort_slow_buckets
                info[r] = st.e.dd_info()
                group.barrier.wait()
                if r == 0 and not args.quiet:
                    hdr = st.x3r.cpu().numpy().view(np.int32).reshape(P, -1)[:, 0]
                    log(f"   pieces per rank {hdr.tolist()}")
                    log(f"step {s}: n_loc={[x.n_loc for x in steppers]} stride={st.stride} "
                        f"let={st.let_counts.tolist()} retries={st.let_retries} emig_max={st.mig_last} "
                        f"emig_per_rank={[i[1] for i in info]} boundaries_moved={info[0][2]} (this step: "
                        f"{('kept', 'exact quantiles', 'sample quantiles')[info[0][3]]}) "
                        f"x2_received_bytes={P * (32 + 32 * st.mig_stride_used)} "
                        f"mig_rounds={st.mig_rounds} flags={flags} slow_buckets={slow} "
                        f"{(time.time() - t0) * 1e3:.1f} ms")
                group.barrier.wait()
        except BaseException as ex:  # noqa: BLE001
            errs.append((r, ex))
            log(f"rank {r}: {ex!r}")
            group.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise SystemExit(1)
    if args.check:
        pos = np.zeros((n, 3), np.float32)
        acc = np.zeros((n, 3), np.float32)
        for st in steppers:
            ids, posm, vel, a = st.local_state()
            pos[ids] = posm[:, :3]
            acc[ids] = a
        with pkg.Engine(n, theta=args.theta) as e:
            e.upload(*ic)
            e.step(args.steps)
            x, y, z, *_ = e.download()
            ax, ay, az = e.download_acc()
        p1 = np.stack([x, y, z], 1)
        a1 = np.stack([ax, ay, az], 1)
        rel = np.linalg.norm(acc - a1, axis=1) / np.maximum(np.linalg.norm(a1, axis=1), 1e-30)
        bad = rel.max() > 5e-3 or np.median(rel) > 1e-5
        log(f"{'FAIL' if bad else 'ok  '} world {P} n {n} seed {args.seed} theta {args.theta} {args.ic}: max |dpos| {np.abs(pos - p1).max():.3e}  acc rel err median {np.median(rel):.3e} "
            f"p99.99 {np.quantile(rel, 0.9999):.3e} max {rel.max():.3e}")


if __name__ == "__main__":
    main()
