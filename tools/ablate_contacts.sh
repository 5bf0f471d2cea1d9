# Times bench.py with parts of k_contacts skipped (HT_DEBUG_SKIP bits 8 = no pairs, 32 = no polytope); results are NOT valid poses.
# Needs a tuning build: HT_TUNING=1 python -m hand_tracking_samples_amd.build --force (rebuild without it afterwards).
export HT_BENCH_TUNING_RUN=1
mkdir -p gpurun_out; rm -f gpurun_out/ablc.log
for d in 0 32 8; do
  HT_DEBUG_SKIP=$d python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dbg',$d,'ms/step',j['ms_per_step'],'contacts ms/step',j['phase_ms_per_step'].get('contacts'),'solve',j['phase_ms_per_step'].get('solve'))
" >> gpurun_out/ablc.log || exit 1
done
cat gpurun_out/ablc.log
