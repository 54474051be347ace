#!/bin/bash
# dd_mark_kernel's duration under rocprofv3 for library builds, same box: bh_bench --replay-rank 3 of an 8 x 1M rehearsal
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset LD_PRELOAD; else export LD_PRELOAD=$R/tools/bin/libs/$v.so; fi
  rm -rf $O/prof_mark
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mark -- ./nbody-barnes-hut-cuda_amd/bh_bench --n 8000000 --ic plummer --devices 0,0,0,0,0,0,0,0 --steps 6 --warmup 4 --quiet --replay-rank 3 > $O/mark_ab_$v.txt 2>&1
  unset LD_PRELOAD
  f=$(find $O/prof_mark -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "dd_x1_pack_kernel" in r["Kernel_Name"])
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[last:] if "dd_mark_kernel" in r["Kernel_Name"]]
import statistics
print(sys.argv[2], "dd_mark_kernel in the replays of rank 3: n", len(d), "median us", round(statistics.median(d), 2), "min", round(min(d), 2))
PY
done; done
