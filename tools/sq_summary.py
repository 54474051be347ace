"""Summarise the rocprofv3 --pmc passes of tools/sq_force.sh: per-kernel counter means -> <dir>/summary.json,
and the force kernel's counters per (record, wave) on stdout."""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(d, "pass*/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
out = {}
for k, cs in per.items():
    out[k] = {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
    if k in dur:
        out[k]["_duration_ms_under_pmc"] = {"dispatches": len(dur[k]), "mean": sum(dur[k]) / len(dur[k])}
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1, sort_keys=True)
for k in out:
    if "force_fast_kernel" in k:
        print(k[:60])
        for c in sorted(out[k]):
            print(f"  {c:32s} {out[k][c]['mean']:.4g}")
