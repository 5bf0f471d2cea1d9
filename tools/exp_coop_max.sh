export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
for f in 1536 2048 4096 8192; do for c in 1024 100000; do
  HT_CONTACTS_COOP_MAX=$c timeout -k 10 150 python bench.py --frames-per-gpu $f --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $f coop_max $c ms/step',j['ms_per_step'], j.get('verified'))"
done; done
