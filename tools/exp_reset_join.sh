export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
for j in 0 1 2 3 5; do
  HT_RESET_JOIN=$j python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('join',$j,'ms/step',j['ms_per_step'],'verified',j.get('verified'))"
done
