# Produces the round's measurement artefacts under gpurun_out/prof_rNN/ (copy the summaries into profiles/):
#   bench.json                     python bench.py (default: 1024 frames, 20 steps, cpu_baseline)
#   kernel_stats.csv               rocprofv3 --kernel-trace --stats of a short bench run
#   pmc_fetch / pmc_write          two separate counter passes (FETCH_SIZE, WRITE_SIZE), aggregated by tools/pmc_traffic.py
#   pmc_mfma                       one counter pass on the CNN-only workload (MFMA busy cycles), aggregated by tools/pmc_mfma.py
# usage: bash tools/profile_round.sh r01
set -e
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprofv3.json 2> $OUT/trace.err
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv > $OUT/pmc_hbm_traffic.json
head -c 600 $OUT/pmc_hbm_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 bench.py --workload cnn --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma.err
python3 tools/pmc_mfma.py $OUT/pmc_mfma/m_counter_collection.csv > $OUT/pmc_mfma_util.json
