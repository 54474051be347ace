#!/bin/bash
# A/B sweep of force-kernel tunables on the GPU box: prints ms/step and avg force ms
cd $GRAFT_REPO_ROOT
for fb in 256 128 64; do for m in 0 1; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-block $fb --xcd-mode $m > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('force_block $fb xcd_mode $m', round(d['ms_per_step'],3), round(d['stages']['avg_force_ms'],3))"
done; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-block 64 --theta 0.3 > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('theta0.3 force_block 64', round(d['ms_per_step'],3), round(d['stages']['avg_force_ms'],3))"
