#!/bin/bash
# multi-process soak of the partial two-pass step: 2 processes x 500,000 bodies on one GPU over gloo, each with its own
# main and side stream (the launches that share a fold really run at once); tests/dd_gpu_worker.py compares the
# gathered state with a single-context run.   tools/r5_mp_soak.sh [steps=150] [two|one|adaptive]
cd $GRAFT_REPO_ROOT
S=${1:-150}; F=${2:-two}
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 \
  tests/dd_gpu_worker.py gpurun_out/mp_soak.json 1000000 $S $F 0 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*" | tail -5
cat gpurun_out/mp_soak.json; echo
