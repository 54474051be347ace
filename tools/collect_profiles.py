"""Turn the rocprofv3 outputs that tools/profile.sh (and tools/sq_force.sh) left under gpurun_out/ into the
committed summaries of profiles/<tag>/ (newest run of each directory):
python tools/collect_profiles.py r01_v4"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]
out = os.path.join(ROOT, "profiles", tag)
os.makedirs(out, exist_ok=True)


def newest(pattern):
    fs = glob.glob(os.path.join(G, pattern))
    return max(fs, key=os.path.getmtime) if fs else None


ks = newest("prof_trace/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(out, "kernel_stats.csv"))
for src, dst in (("prof_trace.json", "bench_under_rocprof.json"), ("bench_final.json", "bench.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(out, dst))


def counters(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


pmc = {}
for d in ("prof_fetch", "prof_write"):
    f = newest(d + "/*/*counter_collection.csv")
    if not f:
        continue
    for k, cs in counters(f).items():
        for c, v in cs.items():
            pmc.setdefault(k, {})[c] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v)}
json.dump(pmc, open(os.path.join(out, "pmc_fetch_write_per_kernel.json"), "w"), indent=1)

fk = [k for k in pmc if "force_fast_kernel" in k]
if fk:
    k = fk[0]
    cal = {}
    for name in ("keys_kernel", "integrate_kernel"):
        kk = [x for x in pmc if name in x]
        if kk:
            cal[name] = {c: pmc[kk[0]][c]["mean_KiB"] for c in pmc[kk[0]]}
    fetch, write = pmc[k]["FETCH_SIZE"]["mean_KiB"], pmc[k]["WRITE_SIZE"]["mean_KiB"]
    rec = {"n": 1000000, "theta": 0.5, "kernel": "force_fast_kernel", "FETCH_SIZE_KiB": fetch,
           "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "calibration_KiB": cal,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile.sh), mean over the "
                  "force launches of `bench.py --steps 5`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE "
                  "counts 64 B per 128-B request (MI355X_MICROARCH.md HBM section). Calibrated in the same run: keys_kernel "
                  "reads 16 MB and integrate_kernel reads 48 MB (FETCH_SIZE reports half of each); WRITE_SIZE is exact.",
           "profile": tag}
    path = os.path.join(ROOT, "profiles", "force_traffic.json")
    try:
        old = json.load(open(path)).get("records", [])
    except Exception:
        old = []
    old = [r for r in old if r.get("profile") != tag]
    json.dump({"records": old + [rec]}, open(path, "w"), indent=1)
    print("force HBM bytes/launch", rec["hbm_bytes_per_launch"])

sq = {}
for d in ("prof_sq_v0", "prof_sq2_v0"):
    f = newest(d + "/*/*counter_collection.csv")
    if not f:
        continue
    for k, cs in counters(f).items():
        if "force_fast_kernel" in k:
            for c, v in cs.items():
                sq[c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
if sq:
    json.dump({"kernel": "force_fast_kernel", "n": 1000000, "theta": 0.5, "counters": sq},
              open(os.path.join(out, "force_fast_kernel_sq.json"), "w"), indent=1)
    w = 15625 * 3701.0
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM"):
        if c in sq:
            print(c, "per (record, wave) ~", round(sq[c]["mean"] / w, 2))
print(sorted(os.listdir(out)))
