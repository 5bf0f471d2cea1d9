# counter passes for the latency-bound kernels of the step (tools/pmc_solve.py aggregates them): bash tools/pmc_solve.sh <out dir> [bench arguments]
set -e
O=${1:-gpurun_out/pmc_solve}; shift || true
mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o a -- $B > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/p2 -o b -- $B > /dev/null 2> $O/p2.err || rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p2 -o b -- $B > /dev/null 2> $O/p2.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p3 -o c -- $B > /dev/null 2> $O/p3.err
python3 tools/pmc_solve.py $O > $O/pmc_solve_issue.json
find $O -name "*.csv" -size +20M -delete
head -c 1500 $O/pmc_solve_issue.json
