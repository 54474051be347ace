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
for src, dst in (("prof_trace.json", "bench_under_rocprof.json"), ("bench_final.json", "bench.json"),
                 ("step_timeline_1M.txt", "step_timeline_1M.txt"), ("bh_bench_disc500k.txt", "configs/bh_bench_disc500k.txt"),
                 ("bh_bench_disc1m.txt", "configs/bh_bench_disc1m.txt")):
    if os.path.exists(os.path.join(G, src)):
        os.makedirs(os.path.dirname(os.path.join(out, dst)), exist_ok=True)
        shutil.copy(os.path.join(G, src), os.path.join(out, dst))
# (only what the same final_round.sh run left: gpurun_out/ keeps earlier rounds' files of the same names)
t_run = os.path.getmtime(os.path.join(G, "bench_final.json")) - 120 if os.path.exists(os.path.join(G, "bench_final.json")) else 0
for f in glob.glob(os.path.join(G, "cfg_*.json")) + glob.glob(os.path.join(G, "step_timeline_[0-9]*.txt")) + \
        glob.glob(os.path.join(G, "force_trace_*.txt")):
    if os.path.getmtime(f) < t_run:
        continue
    sub = "configs" if os.path.basename(f).startswith("cfg_") else ""
    os.makedirs(os.path.join(out, sub), exist_ok=True)
    shutil.copy(f, os.path.join(out, sub, os.path.basename(f)))


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

# the force launch of bh_step: the FUSED instance (first template argument `true` of force_mixed_kernel /
# force_coop_kernel, last of force_fast_kernel) — the one bench.py's roofline names (round-3 review: the record used to
# come from the un-fused launch of the stage calls); the dispatch with the most traffic if several qualify
def fused(k):
    return ("force_mixed_kernel<true" in k or "force_coop_kernel<true" in k or
            ("force_fast_kernel<" in k and k.split("force_fast_kernel<")[1].split(">")[0].replace(" ", "").endswith("true")))
fk = sorted([k for k in pmc if fused(k) and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]],
            key=lambda k: -pmc[k]["FETCH_SIZE"]["mean_KiB"])
if fk:
    k = fk[0]
    cal = {}
    for name in ("keys_split_kernel", "com_kernel"):
        kk = [x for x in pmc if name in x]
        if kk:
            cal[name] = {c: pmc[kk[0]][c]["mean_KiB"] for c in pmc[kk[0]]}
    fetch, write = pmc[k]["FETCH_SIZE"]["mean_KiB"], pmc[k]["WRITE_SIZE"]["mean_KiB"]
    # which kernel sources the profiled bench ran: its own line says (bench.py build.csrc_sha16); bench.py compares
    # that with the sources it runs and prints roofline.traffic_stale
    sha = None
    for src in ("prof_fetch.json", "prof_write.json"):
        try:
            line = [l for l in open(os.path.join(G, src)) if l.startswith("{")][-1]
            sha = json.loads(line)["build"]["csrc_sha16"]
            break
        except Exception:
            pass
    rec = {"n": 1000000, "theta": 0.5, "kernel": k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
           "fused_with_integrate": True, "dispatches": pmc[k]["FETCH_SIZE"]["dispatches"], "FETCH_SIZE_KiB": fetch,
           "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "calibration_KiB": cal,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile.sh), mean over the "
                  "force launches of `bench.py --steps 5`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE "
                  "counts 64 B per 128-B request (MI355X_MICROARCH.md HBM section). The same table holds the streaming kernels "
                  "of the step as a cross-check (keys_split_kernel reads 16 B and writes 8 B per body, com_kernel<false> "
                  "writes 32 B per record; FETCH_SIZE reports half of what they read, WRITE_SIZE is exact).",
           "profile": tag, "csrc_sha16": sha}
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
