# A/B of the product library against another build on one box: bash tools/ab_lib.sh <other .so> [bench arguments]
OTHER=$1; shift
run() { timeout -k 10 200 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-io --no-two-in-flight $ARGS 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phase_ms_per_step']; print('$1'.split('/')[-1][:40].ljust(40),'frames/s',round(j['value']),'ms/step',j['ms_per_step'],'contacts',p['contacts'],'cloud_rows',p['cloud_rows'],'solve',p['solve'],'verified',j['verified'])"; }
ARGS="$@"
run HT_PRODUCT=1
run HT_LIB_PATH=$PWD/$OTHER HT_BENCH_TUNING_RUN=1
run HT_PRODUCT=1
run HT_LIB_PATH=$PWD/$OTHER HT_BENCH_TUNING_RUN=1
