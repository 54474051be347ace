#!/bin/bash
# round 4: LET marking / export restructured (one work item per candidate x rank, per exporting cell x child slot):
# the DD tests, then kernel time per rank-step of the 8 x 1M and 8 x 125,000 rehearsals
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider > $O/pytest_ddlet.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_ddlet.log
for N in 8000000 1000000; do
  DDFLAGS="--no-split --quiet" tools/dd_profile.sh 8 $N 6 ddlet_$N > $O/ddlet_$N.txt 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$O/prof_ddlet_$N/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "rocclr" not in r["Name"] and "at::native" not in r["Name"]) / 48 / 1e3
print("n_total $N: kernels per rank-step", round(tot, 1), "us (6 steps incl. the first)")
for r in rows:
    if any(k in r["Name"] for k in ("dd_mark", "dd_export_pd_kernel", "dd_let", "force_")): print("   ", r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done
