# Produces the round's measurement artefacts under gpurun_out/prof_rNN/; afterwards `bash tools/profile_round.sh --collect rNN` (no GPU) copies the
# summaries into profiles/rNN_* (the files the README there lists):
#   bench.json                     python bench.py (default: BASELINE configs[2], 1024 frames, 20 steps, cpu_baseline)
#   bench_frames8192.json          one GPU's shard of BASELINE configs[3]
#   bench_cnn.json                 BASELINE configs[1] (CNN forward only)
#   bench_config5.json             BASELINE configs[4] (128x128 frames, 26 bones, the path the reference runs)
#   bench_config5_e2e.json         BASELINE configs[4] end to end (SURVEY 8d config 5 i-iii): the 128x128-input net -> decode -> 26-bone tracker in one unit of work
#   bench_config5_cnn128.json      BASELINE configs[4], the 128x128-input net
#   latency_dropin.jsonl           the drop-in surface: HandTracker::update one frame per call through the C++ driver, ht_update_sync on 8 / 64 trackers (p50 / p99 ms)
#   kernel_stats.csv               rocprofv3 --kernel-trace --stats of a short bench run (1024 frames); kernel_stats_frames8192.csv the same at 8192
#   pmc_fetch / pmc_write          two separate counter passes (FETCH_SIZE, WRITE_SIZE) per workload (1024 frames, 8192 frames, configs[4]), each aggregated
#                                  by tools/pmc_traffic.py into a file that names what it was measured on (bench.py quotes it only for that workload)
#   pmc_mfma, pmc_mfma128          one counter pass each on the CNN-only workloads (MFMA busy cycles), aggregated by tools/pmc_mfma.py
#   afterwards: python tools/cnn_roofline.py profiles/rNN > profiles/rNN_cnn_kernel_roofline.json (per-kernel MFMA roofline of the CNN workloads)
# usage: bash tools/profile_round.sh r02
set -e
if [ "$1" = "--collect" ]; then
	R=${2:-r03}; S=gpurun_out/prof_$R; D=profiles
	for f in bench bench_takecnn pmc_solve_issue pmc_solve_issue_frames8192 bench_frames8192 bench_cnn bench_config5 bench_config5_e2e bench_config5_cnn128 bench_under_rocprofv3 bench_dist1 pmc_hbm_traffic pmc_hbm_traffic_frames8192 pmc_hbm_traffic_config5 pmc_hbm_traffic_config5_e2e pmc_mfma_util pmc_mfma128_util; do [ -f $S/$f.json ] && cp $S/$f.json $D/${R}_$f.json; done
	cp $S/kernel_stats.csv $D/${R}_rocprofv3_kernel_stats.csv; cp $S/kernel_stats_frames8192.csv $D/${R}_rocprofv3_kernel_stats_frames8192.csv
	cp $S/kernel_stats_cnn.csv $D/${R}_rocprofv3_kernel_stats_cnn.csv; cp $S/kernel_stats_cnn128.csv $D/${R}_rocprofv3_kernel_stats_cnn128.csv
	[ -f $S/step_timeline.txt ] && cp $S/step_timeline.txt $D/${R}_step_timeline.txt
	[ -f $S/gpu_tests.log ] && cp $S/gpu_tests.log $D/${R}_gpu_tests.log
	[ -f $S/latency_dropin.jsonl ] && cp $S/latency_dropin.jsonl $D/${R}_latency_dropin.jsonl
	python3 tools/cnn_roofline.py $D/$R > $D/${R}_cnn_kernel_roofline.json
	ls $D/${R}_*
	exit 0
fi
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
python3 bench.py --frames-per-gpu 8192 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_frames8192.json 2> $OUT/bench8192.err
python3 bench.py --workload cnn > $OUT/bench_cnn.json 2> $OUT/bench_cnn.err
python3 bench.py --workload config5 --steps 10 > $OUT/bench_config5.json 2> $OUT/bench_config5.err
python3 bench.py --workload config5-e2e --steps 10 > $OUT/bench_config5_e2e.json 2> $OUT/bench_config5_e2e.err      # BASELINE configs[4] end to end: 128x128 net -> decode -> 26-bone tracker
python3 bench.py --workload config5-cnn128 > $OUT/bench_config5_cnn128.json 2> $OUT/bench_cnn128.err
python3 tools/latency_dropin.py 300 > $OUT/latency_dropin.jsonl 2> $OUT/latency_dropin.err      # HandTracker::update one frame per call, ht_update_sync on 8 / 64 trackers: p50 / p99
python3 bench.py --force-dist --steps 10 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err      # the RCCL gather rehearsed on this box's one rank
[ -n "$SKIP_TESTS" ] || python3 -m pytest tests -m gpu -q -s > $OUT/gpu_tests.log 2>&1 || true      # SKIP_TESTS=1: the suite is run on its own (gpurun's limit per call)
echo "benches done"
# (the legs that put two batches or an upload beside the step are left out: their launches run into each other and would blur the per-kernel averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-host-io > $OUT/bench_under_rocprofv3.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "t_kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/step_timeline.py $(find $OUT/trace -name "t_kernel_trace.csv" | head -1) > $OUT/step_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace8192 -o t -- python3 bench.py --frames-per-gpu 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-two-in-flight --no-host-io > /dev/null 2> $OUT/trace8192.err
cp $(find $OUT/trace8192 -name "t_kernel_stats.csv" | head -1) $OUT/kernel_stats_frames8192.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cnn -o t -- python3 bench.py --workload cnn --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_cnn.err
cp $(find $OUT/trace_cnn -name "t_kernel_stats.csv" | head -1) $OUT/kernel_stats_cnn.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cnn128 -o t -- python3 bench.py --workload config5-cnn128 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_cnn128.err
cp $(find $OUT/trace_cnn128 -name "t_kernel_stats.csv" | head -1) $OUT/kernel_stats_cnn128.csv
echo "traces done"
pmc_pair() {      # name, workload, frames per GPU, extra bench arguments
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $4 > /dev/null 2> $OUT/pmc_fetch.err
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $4 > /dev/null 2> $OUT/pmc_write.err
	python3 tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "f_counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "w_counter_collection.csv" | head -1) $2 $3 > $OUT/$1.json
	rm -rf $OUT/pmc_fetch $OUT/pmc_write
}
pmc_pair pmc_hbm_traffic cnn+solver 1024 ""
pmc_pair pmc_hbm_traffic_frames8192 cnn+solver 8192 "--frames-per-gpu 8192"
pmc_pair pmc_hbm_traffic_config5 config5 1024 "--workload config5"
pmc_pair pmc_hbm_traffic_config5_e2e config5-e2e 1024 "--workload config5-e2e"
head -c 300 $OUT/pmc_hbm_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 bench.py --workload cnn --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma.err
python3 tools/pmc_mfma.py $(find $OUT/pmc_mfma -name "m_counter_collection.csv" | head -1) > $OUT/pmc_mfma_util.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma128 -o m -- python3 bench.py --workload config5-cnn128 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma128.err
python3 tools/pmc_mfma.py $(find $OUT/pmc_mfma128 -name "m_counter_collection.csv" | head -1) > $OUT/pmc_mfma128_util.json
rm -rf $OUT/trace $OUT/trace8192 $OUT/trace_cnn $OUT/trace_cnn128 $OUT/pmc_mfma $OUT/pmc_mfma128
bash tools/pmc_solve.sh $OUT/pmc_solve > /dev/null 2>&1 && cp $OUT/pmc_solve/pmc_solve_issue.json $OUT/pmc_solve_issue.json && rm -rf $OUT/pmc_solve
FRAMES=8192 bash tools/pmc_solve.sh $OUT/pmc_solve8192 --frames-per-gpu 8192 > /dev/null 2>&1 && cp $OUT/pmc_solve8192/pmc_solve_issue.json $OUT/pmc_solve_issue_frames8192.json && rm -rf $OUT/pmc_solve8192      # the same at 8192 frames (two waves per SIMD, records beyond the L2)      # SQ / TCC counters of the latency-bound kernels (issue rate, waits, L2 reads)
python3 bench.py --always-take-cnn --no-cpu-baseline > $OUT/bench_takecnn.json 2> $OUT/bench_takecnn.err      # the headline workload with every CNN-driven pose accepted: verified against poses1024_takecnn.htfx
echo "profile round $R complete"
