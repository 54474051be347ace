#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_r4_15.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_r4_15.log
[ $rc -ne 0 ] && exit 1
python bench.py > $O/bench_r4_15.json 2> $O/bench_r4_15.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_r4_15.json"))
print("1M:", round(d["ms_per_step"],4), "force", round(d["stages"]["avg_force_ms"],4), "frac", round(d["roofline"]["frac"],4), "floorfrac", round(d["roofline"]["issue"]["frac_of_valu_floor"],3))
print(json.dumps(d["roofline"]["issue"]["residency"])[:900])
print("cpu:", json.dumps(d["cpu_baseline"])[:400])
PY
python tools/force_trace.py 1000000 0.5 12 > $O/force_trace_final_1M.txt 2>&1; head -6 $O/force_trace_final_1M.txt
