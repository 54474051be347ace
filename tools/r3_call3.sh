#!/bin/bash
# round 3, GPU call 3: scalar-cache prefetch variants of the force walk
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/pf2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "force" -p no:cacheprovider > gpurun_out/pytest_r3_3.log 2>&1
echo "pytest(pf2) rc=$?"; tail -3 gpurun_out/pytest_r3_3.log
for cfg in "" "--bodies 65536" "--bodies 16384" "--bodies 125000" "--bodies 250000" "--theta 0.3"; do
  echo "== $cfg"; BENCH_ARGS="$cfg" tools/force_ab.sh base pf1 pf2 2>&1 | sort | awk '{a[$1]=a[$1]" "$2"/"$3} END{for(k in a)print k,a[k]}'
done | tee gpurun_out/ab_r3_3.txt
