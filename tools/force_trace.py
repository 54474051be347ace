"""Per-wave timeline of the force launch bh_step makes (bh_force_launch_trace: the traced instance of the same kernel,
grid and placement): start / end of every wave's walk on the chip-wide 100 MHz clock, the SIMD it ran on (HW_ID, XCC_ID).
What it answers: when is the last wave dispatched (T_q), how many waves are resident over time, how long each SIMD
sits idle before the launch ends (the drain), and how much of the launch a perfect redistribution could recover.

    python tools/force_trace.py [n] [theta] [steps]      (BH_LIB_PATH=tools/bin/libs/study.so BH_FORCE_TAIL=0: all one-wave)
Writes gpurun_out/force_trace_<n>_<theta>.npy (raw rows) and prints the summary."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402

pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12

e = pkg.Engine(n, theta=theta)
e.upload(*pkg.plummer(n, seed=42))
e.step(steps)
e.tree_stages()
rows = e.force_launch_trace()   # the launch bh_step makes for this context (bh_force_launch_trace)
W = len(rows)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", f"force_trace_{n}_{theta}.npy"), rows)

t0 = rows[:, 0].astype(np.int64)
t1 = rows[:, 1].astype(np.int64)
base = t0.min()
t0 = (t0 - base) * 0.01  # us
t1 = (t1 - base) * 0.01
hw, xcc = rows[:, 2], rows[:, 3] & 0xF
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0xF  # sh_id + se_id
key = ((xcc.astype(np.int64) * 16 + sh) * 16 + cu) * 4 + simd
span = t1.max()
print(f"n={n} theta={theta} waves={W} span(first start .. last end)={span:.1f} us")
print(f"last wave START (T_q) = {t0.max():.1f} us = {t0.max() / span:.3f} of the span")
life = t1 - t0
print(f"wave lifetime us: mean {life.mean():.1f} p10 {np.percentile(life, 10):.1f} median {np.median(life):.1f} "
      f"p90 {np.percentile(life, 90):.1f} max {life.max():.1f}")
uk, inv = np.unique(key, return_inverse=True)
print(f"distinct SIMDs seen: {len(uk)}; XCDs {len(np.unique(xcc))}")
last_end = np.zeros(len(uk))
np.maximum.at(last_end, inv, t1)
first_start = np.full(len(uk), 1e18)
np.minimum.at(first_start, inv, t0)
cnt = np.bincount(inv)
idle_tail = span - last_end
print(f"waves per SIMD: min {cnt.min()} mean {cnt.mean():.2f} max {cnt.max()}")
print(f"SIMD idle before launch end (us): mean {idle_tail.mean():.1f} median {np.median(idle_tail):.1f} "
      f"p90 {np.percentile(idle_tail, 90):.1f} max {idle_tail.max():.1f}  -> mean/span = {idle_tail.mean() / span:.3f}")
# per XCD
for x in np.unique(xcc):
    m = xcc == x
    print(f"  xcd {x}: waves {m.sum()} last end {t1[m].max():.1f} us, last start {t0[m].max():.1f}")
# resident waves over time
ev = np.concatenate([np.stack([t0, np.ones(W)], 1), np.stack([t1, -np.ones(W)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
res = np.cumsum(ev[:, 1])
tt = ev[:, 0]
area = float(np.sum(res[:-1] * np.diff(tt)))
print(f"mean resident waves over the span: {area / span:.0f} (sum of lifetimes / span)")
for frac in (0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
    k = np.searchsorted(tt, frac * span)
    print(f"  t = {frac:.2f} span: resident {int(res[min(k, len(res) - 1)])}")
# busy SIMDs over time (a SIMD is busy while it holds >= 1 wave): integrate the idle SIMD time inside the span
order = np.argsort(t0, kind="stable")
idle_total = 0.0
for s in range(len(uk)):
    m = inv == s
    a, b = t0[m], t1[m]
    o = np.argsort(a)
    a, b = a[o], b[o]
    cur_end = 0.0
    for x, y in zip(a, b):
        if x > cur_end:
            idle_total += x - cur_end
        cur_end = max(cur_end, y)
    idle_total += span - cur_end
print(f"idle SIMD time inside the span: {idle_total / len(uk):.1f} us per SIMD on average "
      f"({idle_total / len(uk) / span:.3f} of the span)")
# work of a wave under equal sharing of its SIMD: integral over its lifetime of 1 / (waves resident on that SIMD)
work = np.zeros(W)
for s in range(len(uk)):
    idx = np.nonzero(inv == s)[0]
    pts = np.unique(np.concatenate([t0[idx], t1[idx]]))
    for u, v in zip(pts[:-1], pts[1:]):
        act = idx[(t0[idx] <= u) & (t1[idx] >= v)]
        if len(act):
            work[act] += (v - u) / len(act)
print(f"SIMD-time per wave under equal sharing (us): mean {work.mean():.2f} p10 {np.percentile(work, 10):.2f} "
      f"p90 {np.percentile(work, 90):.2f} max {work.max():.2f}; total / SIMDs = {work.sum() / len(uk):.1f} us "
      f"= the launch if every SIMD were busy to the end")
np.save(os.path.join(ROOT, "gpurun_out", f"force_trace_work_{n}_{theta}.npy"), work)
