#!/bin/bash
cd $GRAFT_REPO_ROOT
cp nbody-barnes-hut-cuda_amd/libbh.so /tmp/libbh_keep.so
for rep in 1 2; do for v in "$@"; do
  cp tools/bin/libs/$v.so nbody-barnes-hut-cuda_amd/libbh.so
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), 'build', round(d['stages']['last_step_ms']['build'],4), 'com', round(d['stages']['last_step_ms']['com'],4))"
done; done
cp /tmp/libbh_keep.so nbody-barnes-hut-cuda_amd/libbh.so
