"""Kernel timeline of ONE replayed force phase (bh_bench --replay-rank Q under rocprofv3 --kernel-trace): the replays
follow the run's last dd_x1_pack_kernel, each starts with dd_boxes_kernel; order: form (one pass, 20, 30, 100 %) x
X4 (0, 125, 250 us) x 6 rounds.   python tools/replay_timeline.py <kernel_trace.csv> <form 0..3> <x4 0..2>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
form, x4 = int(sys.argv[2]), int(sys.argv[3])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_step = max(i for i, r in enumerate(rows) if "dd_x1_pack_kernel" in r["Kernel_Name"])
starts = [i for i, r in enumerate(rows) if i > last_step and "dd_boxes_kernel" in r["Kernel_Name"]]
k = (form * 3 + x4) * 6 + 5   # the last round of that cell
i0 = starts[k]
i1 = starts[k + 1] if k + 1 < len(starts) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
print(f"replay {k} of {len(starts)}: form {form} x4 {x4}")
print("   start us     end us   duration  queue  kernel")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:10.2f} {(e - t0) / 1e3:10.2f} {(e - s) / 1e3:9.2f}  q{r['Queue_Id']:>3s}  {name}")
