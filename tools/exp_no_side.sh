# Step time with the side streams off (HT_NO_SIDE: cloud rows, chamber and contacts of a fit step one after the other on the main stream; HT_NO_OVERLAP: the
# error of the carried pose and the reset path not beside the CNN / the first step).  Needs the tuning build (libht_tuning.so).
export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
run() { env "$@" timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${FRAMES:+--frames-per-gpu $FRAMES} 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*','ms/step',j['ms_per_step'])"; }
run HT_X=0
run HT_NO_SIDE=1
run HT_NO_OVERLAP=1
run HT_X=0
