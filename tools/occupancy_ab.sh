#!/bin/bash
# same-box sweep of the force kernel's occupancy: BH_FORCE_LDS bytes of dynamic LDS per (one-wave) workgroup cap the
# waves per CU at 160 KB / bytes (0 = the register limit, 32 waves per CU)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for b in 0 5800 6800 8192 10240; do
  BH_FORCE_LDS=$b python bench.py --steps 40 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds $b', round(d['ms_per_step'],4), round(d['stages']['avg_force_ms'],4))"
done; done
