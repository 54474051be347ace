#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 120 ./tools/bin/ubench_occ > $O/ubench_occ.txt 2>&1; cat $O/ubench_occ.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_r4_5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_r4_5.log
python tools/coop_sweep.py 16384 65536 125000 300000 > $O/coop_sweep_2.txt 2>&1; cat $O/coop_sweep_2.txt
for rep in 1 2 3; do
for n in 1000000; do
  python bench.py --bodies $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product n=$n', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), 'floorfrac', round(d['roofline']['issue']['frac_of_valu_floor'],3))"
  for T in 0 2560 3072 3584; do
    BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so BH_FORCE_TAIL=$T python bench.py --bodies $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$n tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  done
done
done
for T in 0 3072; do for L in 0 8192 10240; do
    BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so BH_FORCE_TAIL=0 BH_FORCE_LDS=$L python bench.py --bodies 1000000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds cap $L', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done; break; done
