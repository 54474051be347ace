#!/bin/bash
# round 3: GPU suite + the bench configurations in one call.  tools/check_round.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-chk}; O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_$T.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_$T.log
python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_$T.json"))
print("1M:", round(d["ms_per_step"],4), "force", round(d["stages"]["avg_force_ms"],4), {k: round(v,4) for k,v in d["stages"]["last_step_ms"].items()}, "frac", round(d["roofline"]["frac"],4), "lane", round(d["roofline"]["lane_efficiency"],3), "floorfrac", round(d["roofline"]["issue"]["frac_of_valu_floor"],3))
print("cpu:", json.dumps(d["cpu_baseline"])[:900])
PY
for cfg in "65536 0.5" "125000 0.5" "500000 0.5" "1000000 0.3"; do
  set -- $cfg
  python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/cfg_${T}_$1_$2.json
  python -c "
import json,sys; d=json.loads(open('$O/cfg_${T}_$1_$2.json').read()); print('$1', '$2', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), {k: round(v,4) for k,v in d['stages']['last_step_ms'].items()}, 'floorfrac', round(d['roofline']['issue']['frac_of_valu_floor'],3))"
done
./nbody-barnes-hut-cuda_amd/bh_bench --n 500000 --steps 200 --warmup 20 --quiet > $O/bh_bench_${T}_disc500k.txt 2>&1; tail -3 $O/bh_bench_${T}_disc500k.txt
