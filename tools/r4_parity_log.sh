#!/bin/bash
# round 4: the GPU suite with its printed distributions (-> profiles/r04_parity/)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > $O/pytest_r4_parity.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_r4_parity.log
grep -E "K=|golden|strict \|da\||fast rel|8 ranks x 1M vs|coop vs one-wave|boundaries moved|fast vs|largest X4" $O/pytest_r4_parity.log
