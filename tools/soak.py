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
    # dd P n steps [log_every] [kind]: the domain-decomposed step through the C group API (bh_create_group /
    # bh_step_group: P ranks on this one GPU, in-process transport) — no Python in the step.  log_every = 1 prints every
    # step (bodies / emigrants / boundary action per rank: the step log of profiles/r05_dd/).
    import ctypes as C
    from nbody_barnes_hut_cuda_amd import _lib as L
    P, n, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    every = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    kind = sys.argv[6] if len(sys.argv) > 6 else "plummer"   # plummer | disc | stream (two halves passing through each
    ic = [a.copy() for a in (pkg.disc if kind == "disc" else pkg.plummer)(n, seed=17)]   # other) | cold (collapse from rest)
    if kind == "stream":
        ic[3] += 400.0
        ic[3][: n // 2] -= 800.0
    if kind == "cold":
        ic[3][:] = 0.0; ic[4][:] = 0.0; ic[5][:] = 0.0
    g = C.c_void_p()
    dev = (C.c_int * P)(*([0] * P))
    st = L.lib.bh_create_group(C.byref(g), P, dev, n, None, None, 0)
    assert st == 0, st
    assert L.lib.bh_group_upload(g, *[np.ascontiguousarray(a).ctypes.data_as(L._F) for a in ic]) == 0
    t0 = time.time()
    moved_prev = 0
    for s in range(0, steps, every):
        k = min(every, steps - s)
        st = L.lib.bh_step_group(g, k)
        info = [L.BhRankInfo() for _ in range(P)]
        dd = np.zeros((P, 8), np.int32)
        flags = []
        for q in range(P):
            r = L.lib.bh_group_rank(g, q)
            L.lib.bh_rank_get_info(r, C.byref(info[q]))
            ctx = L.lib.bh_rank_ctx(r)
            L.lib.bh_dd_get_info(ctx, dd[q].ctypes.data_as(C.POINTER(C.c_int32)))
            bs = L.BhStats(); L.lib.bh_get_stats(ctx, C.byref(bs)); flags.append(bs.status_flags)
        line = (f"step {s + k}: bodies {[i.n_loc for i in info]} emigrants found {dd[:, 1].tolist()} "
                f"boundaries {['kept', 'moved to the proposed quantiles', 'sample quantiles'][int(dd[0, 3])]} "
                f"(moved in {int(dd[0, 2])} steps so far) X2 slots {info[0].mig_stride} LET {list(info[0].let_counts[:P])} "
                f"stride {info[0].stride} extra rounds {info[0].mig_rounds} LET retries {info[0].let_retries}")
        if st != 0 or any(flags):
            print(line, flush=True)
            print(f"FAILED: status {st} ({L.lib.bh_strerror(st).decode()}), left_rank {info[0].left_rank}, flags {flags}", flush=True)
            raise SystemExit(1)
        if every == 1 or (s // every) % 5 == 0 or s + k == steps:
            print(line, flush=True)
    tot = sum(i.n_loc for i in info)
    assert tot == n, tot
    out = [np.full(n, np.nan, np.float32) for _ in range(6)]
    assert L.lib.bh_group_download(g, *[a.ctypes.data_as(L._F) for a in out]) == 0
    assert all(np.isfinite(a).all() for a in out)
    print(f"dd {P} ranks x {n // P} x {steps} steps ({kind}) ok: boundaries moved in {int(dd[0, 2])} of {steps} steps, "
          f"{time.time() - t0:.1f} s")
    L.lib.bh_destroy_group(g)
