#!/bin/bash
# round 3, GPU call 2: GPU suite, force A/B (1M theta 0.5 / 0.3, 65,536), kernel timeline at 65,536 bodies
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/pytest_r3_2.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_r3_2.log
grep -h "p50\|p99" gpurun_out/pytest_r3_2.log | head -40
tools/force_ab.sh base prev > gpurun_out/ab_r3_2.txt 2>&1; cat gpurun_out/ab_r3_2.txt
BENCH_ARGS="--theta 0.3" tools/force_ab.sh base prev > gpurun_out/ab_r3_2_t03.txt 2>&1; cat gpurun_out/ab_r3_2_t03.txt
BENCH_ARGS="--bodies 65536" tools/force_ab.sh base prev > gpurun_out/ab_r3_2_64k.txt 2>&1; cat gpurun_out/ab_r3_2_64k.txt
tools/ktrace.sh n64k --bodies 65536 > gpurun_out/kt_n64k.txt 2>&1; cat gpurun_out/kt_n64k.txt
cp gpurun_out/kt_n64k/*/*kernel_trace.csv gpurun_out/kt_n64k_trace.csv
