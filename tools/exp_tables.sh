# Round 6 A/B on one box (tuning build): the solve tables made by k_solve_prep (HT_TABLES) against k_solve's own prologue (default), each with the row producers of a
# fit step beside the contact kernel (default) and in order on one stream (HT_NO_SIDE).
export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_tuning.so HT_BENCH_TUNING_RUN=1
run() { env "$@" timeout -k 10 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-io --no-two-in-flight ${FRAMES:+--frames-per-gpu $FRAMES} $ARGS 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*','ms/step',j['ms_per_step'],'k_solve',j['roofline']['avg_launch_ms'])"; }
run HT_X=0
run HT_TABLES=1
run HT_NO_SIDE=1
run HT_NO_SIDE=1 HT_TABLES=1
run HT_X=0
