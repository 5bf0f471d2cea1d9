export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
run() { env "$@" timeout -k 10 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-io 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*','ms/step',j['ms_per_step'],'cloud',j['phase_ms_per_step'].get('cloud_rows'))"; }
run HT_X=0
run HT_CLOUD_SPLIT=1
run HT_CLOUD_SPLIT=3
run HT_CLOUD_SPLIT=4
run HT_X=0
