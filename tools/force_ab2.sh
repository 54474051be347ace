#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
  if [ $v = 1 ]; then export BH_STACK1=1; else unset BH_STACK1; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>gpurun_out/ab2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stack1=$v', round(d['ms_per_step'],4), round(d['stages']['avg_force_ms'],4))"
done; done; tail -2 gpurun_out/ab2.err
