#!/bin/bash
# round 4: the studies behind DESIGN.md §4 "Round 4" in one GPU call (outputs: gpurun_out/r4_*.txt -> profiles/r04_*/)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
S=$GRAFT_REPO_ROOT/tools/bin/libs/study.so   # tools/mkvariant.sh study -DBH_STUDY
./tools/bin/ubench_occ > $O/r4_ubench_occ.txt 2>&1; cat $O/r4_ubench_occ.txt
python tools/walk_probe.py > $O/r4_walk_probe.txt 2>&1; grep -v amdgpu $O/r4_walk_probe.txt
python tools/coop_sweep.py 16384 32768 65536 125000 200000 300000 500000 > $O/r4_coop_sweep.txt 2>&1; grep -v amdgpu $O/r4_coop_sweep.txt
# the drain: force ms per number of cooperatively walked groups at the end of the launch (0 = every group by one wave)
for cfg in "1000000 0.5" "1000000 0.3" "500000 0.5" "2000000 0.5"; do
  set -- $cfg
  for T in 0 1024 2048 2389 3072 4096 6144; do
    BH_FORCE_TAIL=$T BH_LIB_PATH=$S python bench.py --bodies $1 --theta $2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$1 theta=$2 tail_groups=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  done
done > $O/r4_drain_sweep.txt 2>&1; cat $O/r4_drain_sweep.txt
BH_FORCE_TAIL=0 BH_LIB_PATH=$S python tools/force_trace.py 1000000 0.5 12 > $O/r4_force_trace_1M_one_wave_per_group.txt 2>&1; head -12 $O/r4_force_trace_1M_one_wave_per_group.txt
for L in 0 8192 10240; do
  BH_FORCE_TAIL=0 BH_FORCE_LDS=$L BH_LIB_PATH=$S python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one wave per group, dynamic LDS per workgroup $L B', 'force', round(d['stages']['avg_force_ms'],4))"
done > $O/r4_occupancy_cap.txt 2>&1; cat $O/r4_occupancy_cap.txt
