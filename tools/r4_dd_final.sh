#!/bin/bash
# round 4: the domain-decomposed step's evidence in one GPU call (-> profiles/r04_dd/)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
bash tools/dd_round.sh r4final
timeout -k 10 300 python tools/dd_debug.py --world 8 --n 8000000 --steps 16 --no-split > $O/dd_step_log_8x1M.txt 2>&1; grep "^step" $O/dd_step_log_8x1M.txt | sed 's/stride=.*emig_max/emig_max/' | cut -c1-260 | tail -8
timeout -k 10 400 bash tools/dd_fuzz.sh > $O/dd_fuzz_r4.txt 2>&1; tail -4 $O/dd_fuzz_r4.txt
mkdir -p $O/soak_r4; timeout -k 10 280 python tools/soak.py dd 8 4000000 300 > $O/soak_r4/dd_8x500k_300.txt 2>&1; tail -2 $O/soak_r4/dd_8x500k_300.txt
