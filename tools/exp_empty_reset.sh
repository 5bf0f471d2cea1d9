# what a masked k_reset launch costs when NO frame is flagged (full_reset_on_error out of reach), 1024 and 8192 frames: rocprofv3 kernel durations
cd /tmp && export TMPDIR=/tmp
for f in 1024 8192; do
  FRAMES=$f rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/er$f -o t -- python3 $GRAFT_REPO_ROOT/tools/time_update.py full_reset_on_error=100 > $GRAFT_REPO_ROOT/gpurun_out/er$f.log 2>&1 || exit 1
  grep -h "k_reset\|k_contacts_coop\|k_cloud_rows" $(find $GRAFT_REPO_ROOT/gpurun_out/er$f -name "t_kernel_stats.csv") | cut -c1-30,150-400
done
