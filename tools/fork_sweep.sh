#!/bin/bash
# where the second stream starts to pay: ms/step with the COM prefix scan riding in the build's launches on one
# stream (library variants with a higher BH_FORK_MIN_N: tools/mkvariant.sh fork400 -DBH_FORK_MIN_N=400000,
# fork1g -DBH_FORK_MIN_N=1000000000) against the product (fork from 163,840 bodies)
cd $GRAFT_REPO_ROOT
for N in 163840 200000 250000 300000 400000 500000 1000000; do
  for L in product fork400 fork1g; do
    if [ $L = product ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=tools/bin/libs/$L.so; fi
    python bench.py --bodies $N --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$N lib=$L ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  done
done
