export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
for d in 0 1048576 0 1048576; do
  HT_DEBUG_SKIP=$d timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $d ms/step',j['ms_per_step'], 'cloud rows phase', j['phase_ms_per_step'].get('cloud_rows'))"
done
