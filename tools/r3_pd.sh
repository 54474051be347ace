#!/bin/bash
# per-destination LET: DD GPU tests, 8 x 1M rehearsal in both X4 flavours (step log + kernel profile), fuzz
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_dist.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest_pd.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest_pd.log; grep "largest X4" $O/pytest_pd.log
for m in 0 1; do
  python tools/dd_debug.py --world 8 --n 8000000 --steps 6 --let-mode $m 2> $O/dd_pd_mode$m.txt
  echo "mode $m:"; grep -o "stride=[0-9]* let=\[[^]]*\] retries=[0-9]*" $O/dd_pd_mode$m.txt | tail -2; grep -o " [0-9.]* ms$" $O/dd_pd_mode$m.txt | tail -3 | tr '\n' ' '; echo
done
bash tools/dd_fuzz.sh > $O/dd_fuzz_pd.txt 2>&1; grep -c "^ok" $O/dd_fuzz_pd.txt; grep "FAIL\|Error" $O/dd_fuzz_pd.txt | head
bash tools/dd_round.sh pd 2>&1 | grep -v "^void\|^(anon\|^__amd" | tail -8
