#!/bin/bash
# rocprofv3 counter passes over the sampling bench (run on the GPU box through gpurun).
# usage: bash tools/pmc_sample.sh <tag>   -> gpurun_out/pmct_<tag>_{a,b,f,w}/ ; read with tools/pmc_show.py
set -e
TAG=${1:-s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmct_${TAG}_a -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmct_${TAG}_b -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/pmct_${TAG}_f -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_f.log 2>&1 || true
echo done
