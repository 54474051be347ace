#!/bin/bash
# round 4, first GPU call: the GPU suite (with its printed distributions), the force-launch traces, a bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > $O/pytest_r4_1.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_r4_1.log
grep -E "K=|golden|strict \|da\||fast rel|8 ranks x 1M vs" $O/pytest_r4_1.log | head -40
export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/ftrace.so
for cfg in "1000000 0.5" "500000 0.5" "1000000 0.3" "2000000 0.5"; do
  set -- $cfg
  timeout -k 10 300 python tools/force_trace.py $1 $2 12 > $O/force_trace_$1_$2.txt 2>&1 || echo "trace $cfg failed"
  head -12 $O/force_trace_$1_$2.txt; tail -4 $O/force_trace_$1_$2.txt
done
unset BH_LIB_PATH
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_r4_1.json 2> $O/bench_r4_1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_r4_1.json"))
print("1M:", round(d["ms_per_step"],4), "force", round(d["stages"]["avg_force_ms"],4), "floorfrac", round(d["roofline"]["issue"]["frac_of_valu_floor"],3))
print(d["roofline"]["issue"]["counted_this_run"])
PY
