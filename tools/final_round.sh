#!/bin/bash
# End-of-round measurement on the GPU box: the default bench line, the rocprofv3 passes behind profiles/<tag>/,
# the other BASELINE configurations, the reference-mode run.   tools/final_round.sh   (then collect_profiles.py <tag>)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -c 600 $O/bench_final.json; echo
tools/profile.sh > $O/profile_sh.log 2>&1
f=$(find $O/prof_trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/step_timeline.py $f 30 > $O/step_timeline_1M.txt && tail -14 $O/step_timeline_1M.txt
for cfg in "16384 0.5" "65536 0.5" "125000 0.5" "500000 0.5" "1000000 0.3" "2000000 0.5" "8000000 0.5"; do
  set -- $cfg
  python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/cfg_$1_$2.json
  python -c "
import json,sys; d=json.loads(open('$O/cfg_$1_$2.json').read()); r=d['roofline']['issue']; print('$1', '$2', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), 'floorfrac', round(r['frac_of_valu_floor'],3), 'resident', round(r['residency']['mean_resident_frac_of_slots'],3) if r.get('residency') else None, {k: round(v,4) for k,v in d['stages']['last_step_ms'].items()})"
done
./nbody-barnes-hut-cuda_amd/bh_bench --n 500000 --steps 200 --warmup 20 --quiet > $O/bh_bench_disc500k.txt 2>&1; tail -4 $O/bh_bench_disc500k.txt
./nbody-barnes-hut-cuda_amd/bh_bench --n 1000000 --steps 200 --warmup 20 --quiet > $O/bh_bench_disc1m.txt 2>&1; tail -4 $O/bh_bench_disc1m.txt
for cfg in "1000000 0.5" "500000 0.5" "65536 0.5"; do
  set -- $cfg
  python tools/force_trace.py $1 $2 12 > $O/force_trace_$1_$2.txt 2>&1; head -3 $O/force_trace_$1_$2.txt
done
