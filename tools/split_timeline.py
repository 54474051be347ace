"""Timeline of ONE steady step of the decomposed step with its two streams (own pass on the side stream beside the LET
kernels and X4) from a rocprofv3 kernel trace: start, end, duration and queue of every kernel between two launches of
dd_x1_pack_kernel.   python tools/split_timeline.py <kernel_trace.csv> [which step, default 20]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "dd_x1_pack_kernel" in r["Kernel_Name"]]
i0, i1 = marks[which], marks[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"step = kernels {i0}..{i1 - 1}")
print("   start us     end us   duration  queue  kernel")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:10.2f} {(e - t0) / 1e3:10.2f} {(e - s) / 1e3:9.2f}  q{r['Queue_Id']:>3s}  {name}")
print(f"step length (x1 pack to x1 pack): {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.2f} us")
