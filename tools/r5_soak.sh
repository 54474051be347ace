#!/bin/bash
# round 5: soak of the final build (profiles/r05_final/soak/): bh_step against stage-by-stage stepping with the radix
# sort, bit-identical states; a single-context long run; the decomposed step through bh_step_group
cd $GRAFT_REPO_ROOT; O=gpurun_out/soak_r5; mkdir -p $O
timeout -k 10 200 python tools/soak.py compare 1000000 2000 100 > $O/compare_1M_2000.txt 2>&1; tail -1 $O/compare_1M_2000.txt
timeout -k 10 200 python tools/soak.py compare 200000 4000 100 > $O/compare_200k_4000.txt 2>&1; tail -1 $O/compare_200k_4000.txt
timeout -k 10 200 python tools/soak.py compare 50000 20000 250 > $O/compare_50k_20000.txt 2>&1; tail -1 $O/compare_50k_20000.txt
BH_SOAK_IC=cold timeout -k 10 200 python tools/soak.py compare 100000 10000 250 > $O/compare_cold_100k_10000.txt 2>&1; tail -1 $O/compare_cold_100k_10000.txt
timeout -k 10 120 python tools/soak.py single 1000000 20000 > $O/single_1M_20000.txt 2>&1; tail -1 $O/single_1M_20000.txt
