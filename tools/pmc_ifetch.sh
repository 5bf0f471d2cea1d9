# instruction-fetch counters of the step's kernels: bash tools/pmc_ifetch.sh <out dir> [bench arguments]
set -e
O=${1:-gpurun_out/pmc_ifetch}; shift || true
mkdir -p $O; export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -i -E "ifetch|icache|SQC_INST|WAIT_INST|INST_LEVEL|SQ_INSTS_BRANCH|SQ_INST_CYCLES" $O/counters.txt | cut -c1-220 > $O/counters_if.txt || true
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-two-in-flight $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -o a -- $B > /dev/null 2> $O/p1.err || echo "p1 failed"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p2 -o b -- $B > /dev/null 2> $O/p2.err || echo "p2 failed"
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("p1", "p2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"])
            if key not in seen: seen.add(key); n[k] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("SQC_ICACHE_REQ", 0)))[:8]:
        print(p, k, n[k], {c: round(v / n[k], 1) for c, v in agg[k].items()})
PY
find $O -name "*.csv" -size +20M -delete
