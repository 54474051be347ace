#!/bin/bash
# same-box sweep of bh_params.xcd_mode (block -> body-chunk placement of the force kernel)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in 0 1 2; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --xcd-mode $m $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd_mode $m', round(d['ms_per_step'],4), round(d['stages']['avg_force_ms'],4))"
done; done
