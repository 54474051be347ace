#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py (GPU box): tools/ktrace.sh <tag> [bench.py args]; prints the per-kernel table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-kt}; shift
OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $* > $OUT/bench.json 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
