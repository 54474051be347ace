#!/bin/bash
# GPU box: the domain-decomposed step against a single context over seeds, rank counts, theta, ICs
cd $GRAFT_REPO_ROOT
for seed in 1 2 3 4 5; do for w in 2 3 5 8; do
  timeout -k 5 120 python tools/dd_debug.py --world $w --n $((w * 20000 + seed * 777)) --steps 6 --seed $seed --check --quiet 2>&1 | grep "ok  \|FAIL\|rank\|Error"
done; done
for th in 0.3 0.7 1.0; do timeout -k 5 120 python tools/dd_debug.py --world 4 --n 90000 --steps 5 --seed 9 --theta $th --check --quiet 2>&1 | grep "ok  \|FAIL\|rank\|Error"; done
timeout -k 5 120 python tools/dd_debug.py --world 6 --n 120000 --steps 5 --seed 4 --ic disc --check --quiet 2>&1 | grep "ok  \|FAIL\|rank\|Error"
timeout -k 5 120 python tools/dd_debug.py --world 4 --n 100000 --steps 8 --seed 8 --stream-ic --check --quiet 2>&1 | grep "ok  \|FAIL\|rank\|Error"
