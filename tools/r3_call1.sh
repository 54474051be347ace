#!/bin/bash
# round 3, GPU call 1: ubench (EXEC halves, SALU cost), GPU test suite with measured tolerances, force A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 60 tools/bin/ubench_exec > gpurun_out/ubench_exec.txt 2>&1 || { echo "ubench failed"; exit 1; }
echo "ubench done"; cat gpurun_out/ubench_exec.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/pytest_r3_1.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_r3_1.log
grep -h "p50\|p99" gpurun_out/pytest_r3_1.log | head -40
tools/force_ab.sh base prev > gpurun_out/ab_r3_1.txt 2>&1
cat gpurun_out/ab_r3_1.txt
