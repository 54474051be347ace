#!/bin/bash
# end-of-round pass: GPU suite, smoke, final_round.sh (bench + profiles + configs), the domain-decomposed round
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_final.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/final_round.sh
bash tools/dd_round.sh fin 2>&1 | grep -v "^void\|^(anon\|^__amd" | tail -4
python tools/dd_debug.py --world 8 --n 8000000 --steps 6 2> gpurun_out/dd_fin_steplog.txt; tail -2 gpurun_out/dd_fin_steplog.txt | cut -c1-200
