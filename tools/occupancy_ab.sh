#!/bin/bash
# same-box sweep of the force kernel's occupancy: BH_FORCE_LDS bytes of dynamic LDS per (one-wave) workgroup cap the
# waves per CU at 160 KB / bytes (0 = the register limit, 32 waves per CU).  The knob exists only in a design-study
# build: tools/mkvariant.sh study -DBH_STUDY, selected here through BH_LIB_PATH
cd $GRAFT_REPO_ROOT
export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so
for rep in 1 2; do for b in 0 5800 6800 8192 10240; do
  BH_FORCE_LDS=$b python bench.py --steps 40 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds $b', round(d['ms_per_step'],4), round(d['stages']['avg_force_ms'],4))"
done; done
