"""design study: timeline of ONE steady-state bh_step from a rocprofv3 kernel trace (…_kernel_trace.csv):
start offset, duration, queue and the idle gap to the previous kernel end on any queue.
  python tools/step_timeline.py <kernel_trace.csv> [n-th force launch to show, default 30]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
f = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("force_fast_kernel", "force_mixed_kernel", "force_coop_kernel"))]
i0, i1 = f[which - 1] + 1, f[which] + 1
t0 = int(rows[i0]["Start_Timestamp"])
end_prev = int(rows[i0 - 1]["End_Timestamp"])
print(f"step = kernels {i0}..{i1 - 1}; previous force ended {(t0 - end_prev) / 1e3:.2f} us before the first start")
last_end = end_prev
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:8.2f}  q{r['Queue_Id']:>3s}  gap {(s - last_end) / 1e3:7.2f}  {name}")
    last_end = max(last_end, e)
print(f"step length (force end to force end): {(int(rows[i1 - 1]['End_Timestamp']) - end_prev) / 1e3:.2f} us")
