cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for pct in 25 50; do
rm -rf $O/prof_pct$pct
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pct$pct -- ./nbody-barnes-hut-cuda_amd/bh_bench --n 8000000 --ic plummer --steps 6 --warmup 2 --devices 0,0,0,0,0,0,0,0 --quiet --split --split-pct $pct > $O/pct$pct.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/prof_pct$pct/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "force_" in r["Name"]: print("pct $pct", r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
