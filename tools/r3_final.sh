#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_final.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/final_round.sh
