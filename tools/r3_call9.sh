#!/bin/bash
# round 3: double-buffered unit walk vs the single-window walk
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "force or steps" -p no:cacheprovider > gpurun_out/pytest_r3_9.log 2>&1
echo "pytest(db) rc=$?"; tail -3 gpurun_out/pytest_r3_9.log
for cfg in "" "--bodies 65536" "--bodies 16384" "--bodies 125000" "--bodies 250000" "--bodies 500000" "--theta 0.3"; do
  echo "== $cfg"; BENCH_ARGS="$cfg" tools/force_ab.sh base nodb 2>&1 | sort | awk '{a[$1]=a[$1]" "$2"/"$3} END{for(k in a)print k,a[k]}'
done | tee gpurun_out/ab_r3_9.txt
