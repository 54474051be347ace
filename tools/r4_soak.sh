#!/bin/bash
# round 4: soak of the final build (profiles/r04_final/soak/): bh_step — cooperative / mixed force launch fused with the
# integrate step, splitter sort, folded cube — against stage-by-stage stepping with the radix sort, bit-identical states
cd $GRAFT_REPO_ROOT; O=gpurun_out/soak_r4; mkdir -p $O
timeout -k 10 280 python tools/soak.py compare 1000000 3000 100 > $O/compare_1M_3000.txt 2>&1; tail -2 $O/compare_1M_3000.txt
timeout -k 10 280 python tools/soak.py compare 400000 4000 100 > $O/compare_400k_4000.txt 2>&1; tail -1 $O/compare_400k_4000.txt
timeout -k 10 280 python tools/soak.py compare 200000 6000 100 > $O/compare_200k_6000.txt 2>&1; tail -1 $O/compare_200k_6000.txt
timeout -k 10 280 python tools/soak.py compare 50000 30000 250 > $O/compare_50k_30000.txt 2>&1; tail -1 $O/compare_50k_30000.txt
BH_SOAK_IC=cold timeout -k 10 280 python tools/soak.py compare 100000 20000 250 > $O/compare_cold_100k_20000.txt 2>&1; tail -1 $O/compare_cold_100k_20000.txt
timeout -k 10 120 python tools/soak.py single 1000000 20000 > $O/single_1M_20000.txt 2>&1; tail -1 $O/single_1M_20000.txt
timeout -k 10 280 python tools/soak.py dd 8 4000000 300 > $O/dd_8x500k_300.txt 2>&1; tail -1 $O/dd_8x500k_300.txt
