cd $GRAFT_REPO_ROOT; O=gpurun_out
DDFLAGS="--quiet" tools/dd_profile.sh 8 8000000 6 ddsplit > $O/ddsplit.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/prof_ddsplit/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if "force_" in r["Name"]: print(r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "rocclr" not in r["Name"] and "at::native" not in r["Name"]) / 48 / 1e3
print("library kernels per rank-step (split mode, sum of durations):", round(tot,1))
PY
