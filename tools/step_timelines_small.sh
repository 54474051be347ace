cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for N in 65536 16384 250000; do
rm -rf $O/prof_tl_$N
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl_$N -- python3 bench.py --bodies $N --steps 40 --warmup 5 --no-cpu-baseline > $O/tl_$N.json 2> $O/tl_$N.err
python tools/step_timeline.py $(find $O/prof_tl_$N -name "*kernel_trace.csv" | head -1) 30 > $O/step_timeline_$N.txt; cat $O/step_timeline_$N.txt
done
