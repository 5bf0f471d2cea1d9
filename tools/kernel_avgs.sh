# average kernel durations of a short bench run under rocprofv3 (1024 frames): bash tools/kernel_avgs.sh [workload, default cnn+solver]
WL=${1:-cnn+solver}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kavg
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kavg -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/kavg.err || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kavg/**/t_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0]
    if float(r["Percentage"])>0.5: print("%-40s %4s calls  avg %8.1f us  %5s %%" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
