"""Long-run soak: many steps, device-side status flags and finiteness checked along the way.
python tools/soak.py single 1000000 3000 | python tools/soak.py dd 4 1000000 600
python tools/soak.py compare 200000 3000 [every]: bh_step with all its shortcuts (cube folded by the previous integrate,
splitter sort, last-block hand-offs) against a context stepped stage by stage with the radix sort; the states must
stay bit-identical (a rare race in a hand-off would show as a mismatch or a sticky flag)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg
pkg = bhpkg.load()
mode = sys.argv[1]
if mode == "compare":
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    every = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    ic = pkg.plummer(n, seed=23)
    if os.environ.get("BH_SOAK_IC") == "cold":  # the same bodies at rest: they fall into a dense core (close pairs,
        ic = ic[:3] + tuple(np.zeros_like(v) for v in ic[3:6]) + ic[6:]  # equal keys, deep trees, big cube changes)
    a = pkg.Engine(n)
    b = pkg.Engine(n, sort_variant=2)
    a.upload(*ic); b.upload(*ic)
    t0 = time.time()
    for s in range(0, steps, every):
        a.step(every)
        for _ in range(every):
            b.bbox(); b.morton(); b.sort(); b.build(); b.com(); b.force(); b.integrate()
        sa, sb = np.stack(a.download(), 1), np.stack(b.download(), 1)
        assert a.stats().status_flags == 0 and b.stats().status_flags == 0, (s, a.stats().status_flags)
        assert sa.tobytes() == sb.tobytes(), f"states differ after step {s + every}"
        assert np.array_equal(a.download_order(), b.download_order())
        if (s // every) % 10 == 0:
            print(f"step {s + every}: identical, cells {a.stats().n_internal}", flush=True)
    print(f"compare {n} x {steps} steps ok, {time.time() - t0:.1f} s")
elif mode == "single":
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    e = pkg.Engine(n)
    e.upload(*pkg.plummer(n, seed=17))
    t0 = time.time()
    for s in range(0, steps, 250):
        e.step(min(250, steps - s))
        st = e.stats()
        assert st.status_flags == 0, (s, st.status_flags)
        print(f"step {s + 250}: flags 0, cells {st.n_internal}, max level {st.max_level}", flush=True)
    x, y, z, vx, vy, vz = e.download()
    assert np.isfinite(x).all() and np.isfinite(vx).all()
    print(f"single {n} x {steps} steps ok, {time.time() - t0:.1f} s, |x|max {np.abs(x).max():.1f}")
else:
    import torch
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    P, n, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    kind = sys.argv[6] if len(sys.argv) > 6 else "plummer"   # plummer | disc | stream (two halves passing through each
    ic = [a.copy() for a in (pkg.disc if kind == "disc" else pkg.plummer)(n, seed=17)]   # other) | cold (collapse from rest)
    if kind == "stream":
        ic[3] += 400.0
        ic[3][: n // 2] -= 800.0
    if kind == "cold":
        ic[3][:] = 0.0; ic[4][:] = 0.0; ic[5][:] = 0.0
    ic = tuple(ic)
    order = bhdist.global_morton_order(pkg, ic, 0)
    group = bhdist.LocalGroup(P)
    stream = torch.cuda.Stream(0)
    errs, sts = [], [None] * P
    log_from = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 30
    hist = [[] for _ in range(P)]
    def work(r):
        try:
            torch.cuda.set_device(0)
            st = bhdist.DomainStepper(pkg, ic, bhdist.LocalComm(group, r), 0, stream=stream, order=order)
            sts[r] = st
            group.barrier.wait()
            for s in range(0, steps, 100):
                if s >= log_from:   # (argv[5]: log every step from here on; printed when the run fails)
                    for k in range(min(100, steps - s)):
                        st.step(1)
                        hist[r].append((s + k + 1, st.n_loc, st.e.dd_info(), st.mig_last, st.mig_stride_used, st.mig_rounds))
                        del hist[r][:-14]
                else:
                    st.step(min(100, steps - s))
                fl = st.e.stats().status_flags
                assert fl == 0, (r, s, fl)
                group.barrier.wait()
                if r == 0:
                    print(f"step {s + 100}: n_loc {[x.n_loc for x in sts]} let {st.let_counts.tolist()} emig {st.mig_last} "
                          f"mig_rounds {st.mig_rounds} let_retries {st.let_retries}", flush=True)
                group.barrier.wait()
        except BaseException as ex:
            errs.append((r, ex)); print("rank", r, repr(ex), flush=True); group.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]; [t.join() for t in th]
    if errs:
        for r in range(P):
            for h in hist[r]:
                print(f"rank {r} step {h[0]}: n_loc {h[1]} info(bodies, emigrants, moves, mode) {h[2][:4]} emig_max {h[3]} stride {h[4]} rounds {h[5]}")
        raise SystemExit(1)
    tot = sum(s.n_loc for s in sts)
    assert tot == n, tot
    print(f"dd {P} ranks x {n // P} x {steps} steps ({kind}) ok")
