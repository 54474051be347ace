#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dd.py -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_link.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_link.log
for cfg in "" "--theta 0.3" "--bodies 65536" "--bodies 500000"; do
  echo "== $cfg"; BENCH_ARGS="$cfg" tools/force_ab.sh base prev 2>&1 | sort | awk '{a[$1]=a[$1]" "$2"/"$3} END{for(k in a)print k,a[k]}'
done | tee gpurun_out/ab_link.txt
