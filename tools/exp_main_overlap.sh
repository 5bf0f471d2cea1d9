# Main-pass organisation experiment (needs the tuning build: HT_TUNING=1 python -m hand_tracking_samples_amd.build --force --out libht_tuning.so):
# contacts beside the cloud rows as the lane-per-pair kernel, and the cloud rows' blocks per frame.
export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${FRAMES:+--frames-per-gpu $FRAMES} 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*','ms/step',j['ms_per_step'],'verified',j.get('verified'))"
}
run HT_X=0
run HT_CONTACTS_MAIN_LANES=1
run HT_CLOUD_SPLIT=1
run HT_CLOUD_SPLIT=3
run HT_CONTACTS_MAIN_LANES=1 HT_CLOUD_SPLIT=1
run HT_CONTACTS_MAIN_LANES=1 HT_CLOUD_SPLIT=3
