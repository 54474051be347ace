#!/bin/bash
# A/B sweep of force-kernel tunables on the GPU box: prints ms/step and avg force ms
cd $GRAFT_REPO_ROOT
for v in 1 0; do for m in 0 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --variant $v --xcd-mode $m > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('variant $v xcd_mode $m', round(d['ms_per_step'],3), round(d['stages']['avg_force_ms'],3))"
done; done
for cap in 4 8 16; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --variant 1 --xcd-mode 0 --leaf-cap $cap > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('variant 1 leaf_cap $cap', round(d['ms_per_step'],3), round(d['stages']['avg_force_ms'],3), d['stages']['last_step_ms']['build'], d['roofline']['per_body'])"
done
