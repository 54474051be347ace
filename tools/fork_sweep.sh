#!/bin/bash
# where the second stream starts to pay: ms/step with the COM prefix scan riding in the build's launches on one
# stream (library variant fork1g: tools/mkvariant.sh fork1g -DBH_FORK_MIN_N=1000000000) against the second stream from
# 163,840 bodies (fork163 -DBH_FORK_MIN_N=163840)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for N in 65536 163840 250000 500000 700000 1000000 1500000 2000000 4000000; do
  for L in fork163 fork1g; do
    BH_LIB_PATH=tools/bin/libs/$L.so python bench.py --bodies $N --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$N lib=$L ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), 'rest', round(d['ms_per_step']-d['stages']['avg_force_ms'],4))"
  done
done; done
