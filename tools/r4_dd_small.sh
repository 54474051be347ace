#!/bin/bash
# round 4: the passes of the domain-decomposed step as cooperative launches at the strong-scaling sizes
# (1M bodies over 8 ranks = 125,000 per rank; 500k over 8): kernel time per rank-step, one wave per group against K waves
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider > $O/pytest_ddsmall.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_ddsmall.log
for N in 1000000 500000; do for C in 1 0; do
  DDFLAGS="--no-split --quiet --force-coop $C" tools/dd_profile.sh 8 $N 8 ddsm_${N}_$C > $O/ddsm_${N}_$C.txt 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$O/prof_ddsm_${N}_$C/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 64 / 1e3
force = sum(float(r["TotalDurationNs"]) for r in rows if "force_" in r["Name"]) / 64 / 1e3
print("n_total $N force_coop $C: kernels per rank-step", round(tot, 1), "us, of which force", round(force, 1))
for r in rows:
    if "force_" in r["Name"]: print("   ", r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done; done
