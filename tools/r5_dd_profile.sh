#!/bin/bash
# round 5: kernel statistics of the domain-decomposed step rehearsed on ONE GPU through the C++ host
# (bh_bench --devices 0,0,...: bh_create_group / bh_step_group, in-process transport), one pass and two passes.
#   tools/r5_dd_profile.sh [ranks=8] [n_total=8000000] [steps=6] [tag=r5dd]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
P=${1:-8}; N=${2:-8000000}; S=${3:-6}; TAG=${4:-r5dd}
DEV=$(python3 -c "print(','.join(['0']*$P))")
cd $R
for mode in onepass split; do
  FLAG="--one-pass"; [ $mode = split ] && FLAG="--split"
  rm -rf $O/prof_${TAG}_$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$mode -- \
    ./nbody-barnes-hut-cuda_amd/bh_bench --n $N --ic plummer --steps $S --warmup 2 --devices $DEV --quiet $FLAG \
    > $O/${TAG}_$mode.txt 2>&1 || { echo "run failed ($mode)"; tail -5 $O/${TAG}_$mode.txt; exit 1; }
  tail -2 $O/${TAG}_$mode.txt
  python3 - <<PY
import csv, glob
f = glob.glob("$O/prof_${TAG}_$mode/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = 2 + 2 * $S + 5          # warm-up, the frames twice (with and without a sync per frame), the profiled steps
rs = steps * $P
out = open("$O/${TAG}_${mode}_table.txt", "w")
print("$P ranks x %d bodies, bh_bench --devices (in-process transport), $mode; per-kernel total / %d rank-steps" % ($N // $P, rs), file=out)
tot = lib = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"]) / rs / 1e3
    tot += t
    if "rocclr" not in r["Name"]: lib += t
    print("%-72s calls %6s per rank-step %8.1f us avg %8.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:72], r["Calls"], t, float(r["AverageNs"]) / 1e3), file=out)
print("total per rank-step us %.1f; the library's own kernels %.1f" % (tot, lib), file=out)
out.close()
print(open("$O/${TAG}_${mode}_table.txt").read()[:2600])
PY
done
