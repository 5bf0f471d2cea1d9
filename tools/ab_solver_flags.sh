set -e
mkdir -p gpurun_out/r2d
for v in "slp|-mllvm -amdgpu-sched-strategy=max-ilp" "noslp|-mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize" "default|" "noslp_only|-fno-slp-vectorize"; do
  name=${v%%|*}; flags=${v#*|}
  HT_TUNING=1 HT_SOLVER_FLAGS="$flags" python -m hand_tracking_samples_amd.build --force > /dev/null 2>&1
  bash tools/ablate_solve.sh > /dev/null 2>&1
  echo "== $name ($flags)"; cat gpurun_out/abl.log
done
