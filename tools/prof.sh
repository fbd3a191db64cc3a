#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box through gpurun).
# usage: bash tools/prof.sh <tag>   -> gpurun_out/prof_<tag>_{stats,pmc1,pmc2}/
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="--no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0"  # sampling leg only: the region the roofline figures are computed on
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python3 bench.py --steps 10 --warmup 3 $ARGS > gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${TAG}_pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-graph $ARGS > gpurun_out/prof_${TAG}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/prof_${TAG}_pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-graph $ARGS > gpurun_out/prof_${TAG}_pmc2.log 2>&1
find gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_pmc1 gpurun_out/prof_${TAG}_pmc2 -name "*.csv" | head -20
# HBM traffic counters, each in its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmc3 -- python3 bench.py --steps 3 --warmup 1 --no-graph $ARGS > gpurun_out/prof_${TAG}_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmc4 -- python3 bench.py --steps 3 --warmup 1 --no-graph $ARGS > gpurun_out/prof_${TAG}_pmc4.log 2>&1
