# Produces the round's measurement artefacts under gpurun_out/prof_rNN/ (copy the summaries into profiles/ with tools/collect_profiles.sh):
#   bench.json                     python bench.py (default: BASELINE configs[2], 1024 frames, 20 steps, cpu_baseline)
#   bench_frames8192.json          one GPU's shard of BASELINE configs[3]
#   bench_cnn.json                 BASELINE configs[1] (CNN forward only)
#   bench_config5.json             BASELINE configs[4] (128x128 frames, 26 bones, the path the reference runs)
#   bench_config5_cnn128.json      BASELINE configs[4], the 128x128-input net
#   kernel_stats.csv               rocprofv3 --kernel-trace --stats of a short bench run (1024 frames); kernel_stats_frames8192.csv the same at 8192
#   pmc_fetch / pmc_write          two separate counter passes (FETCH_SIZE, WRITE_SIZE), aggregated by tools/pmc_traffic.py
#   pmc_mfma, pmc_mfma128          one counter pass each on the CNN-only workloads (MFMA busy cycles), aggregated by tools/pmc_mfma.py
#   afterwards: python tools/cnn_roofline.py profiles/rNN > profiles/rNN_cnn_kernel_roofline.json (per-kernel MFMA roofline of the CNN workloads)
# usage: bash tools/profile_round.sh r02
set -e
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
python3 bench.py --frames-per-gpu 8192 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_frames8192.json 2> $OUT/bench8192.err
python3 bench.py --workload cnn > $OUT/bench_cnn.json 2> $OUT/bench_cnn.err
python3 bench.py --workload config5 --steps 10 > $OUT/bench_config5.json 2> $OUT/bench_config5.err
python3 bench.py --workload config5-cnn128 > $OUT/bench_config5_cnn128.json 2> $OUT/bench_cnn128.err
echo "benches done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprofv3.json 2> $OUT/trace.err
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace8192 -o t -- python3 bench.py --frames-per-gpu 8192 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/trace8192.err
cp $OUT/trace8192/t_kernel_stats.csv $OUT/kernel_stats_frames8192.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cnn -o t -- python3 bench.py --workload cnn --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_cnn.err
cp $OUT/trace_cnn/t_kernel_stats.csv $OUT/kernel_stats_cnn.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cnn128 -o t -- python3 bench.py --workload config5-cnn128 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_cnn128.err
cp $OUT/trace_cnn128/t_kernel_stats.csv $OUT/kernel_stats_cnn128.csv
echo "traces done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv > $OUT/pmc_hbm_traffic.json
head -c 300 $OUT/pmc_hbm_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 bench.py --workload cnn --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma.err
python3 tools/pmc_mfma.py $OUT/pmc_mfma/m_counter_collection.csv > $OUT/pmc_mfma_util.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma128 -o m -- python3 bench.py --workload config5-cnn128 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma128.err
python3 tools/pmc_mfma.py $OUT/pmc_mfma128/m_counter_collection.csv > $OUT/pmc_mfma128_util.json
rm -rf $OUT/trace $OUT/trace8192 $OUT/trace_cnn $OUT/trace_cnn128 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_mfma128
echo "profile round $R complete"
