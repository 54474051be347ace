#!/bin/bash
# round 3: LDS bucket capacity 12288 vs 8192 (single GPU sort time; slow buckets in the 8 x 1M rehearsal), integrate / keys tile variants
cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in base ls12k intsmall kssmall; do
  if [ "$v" = base ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so; fi
  for rep in 1 2; do
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), {k: round(x,4) for k,x in d['stages']['last_step_ms'].items()})"
  done
done
for v in base ls12k; do
  if [ "$v" = base ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so; fi
  python tools/dd_debug.py --world 8 --n 8000000 --steps 8 2> $O/dd_slow_$v.txt; echo "$v:"; grep -o "slow_buckets=\[[^]]*\]" $O/dd_slow_$v.txt | tail -3; grep -o " [0-9.]* ms$" $O/dd_slow_$v.txt | tail -4 | tr '\n' ' '; echo
done
