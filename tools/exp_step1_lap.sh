# The reset frames' first step beside the batch's preparation of its second (run_update) against the plain order, at 1024 and 8192 frames.  Needs the tuning build (libht_tuning.so).
export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
for f in 1024 8192; do for lap in 0 1; do
  if [ $lap = 1 ]; then export HT_NO_STEP1_LAP=1; else unset HT_NO_STEP1_LAP; fi
  python bench.py --frames-per-gpu $f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames',$f,'plain order' if $lap else 'lapped','ms/step',j['ms_per_step'],'verified',j.get('verified'))"
done; done
