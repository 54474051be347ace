#!/bin/bash
# round 3: domain-decomposed step: tests, 8 x 1M in-process rehearsal profile, world-1 RCCL step
cd $GRAFT_REPO_ROOT
T=${1:-dd1}; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_dist.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest_$T.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_$T.log
DDFLAGS="--no-split --quiet" tools/dd_profile.sh 8 8000000 6 ${T}ns > $O/ddprof_$T.txt 2>&1; tail -3 $O/ddprof_$T.txt
python - <<PY
import csv, glob
f = glob.glob("$O/prof_${T}ns/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"]) / 48 / 1e3; tot += t
    if t > 4: print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):5d} per rank-step {t:8.1f} us avg {float(r["AverageNs"])/1e3:8.1f}')
print("total per rank-step us", round(tot, 1))
import shutil; shutil.copy(f, "$O/dd_kernel_stats_$T.csv")
PY
BH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_${T}_world1.json 2> $O/bench_${T}_world1.err
python -c "
import json; d=json.load(open('$O/bench_${T}_world1.json')); print('world-1 RCCL step ms', round(d['ms_per_step'],4), d['config'].get('domain',{}).get('phase_ms_rank0'))"
