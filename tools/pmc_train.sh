#!/bin/bash
# rocprofv3 counter passes over the training bench (run on the GPU box through gpurun).
# usage: bash tools/pmc_train.sh <tag>   -> gpurun_out/pmct_<tag>_{a,b,f,w}/ ; read with tools/pmc_show.py
set -e
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="--mode train --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmct_${TAG}_a -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmct_${TAG}_b -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmct_${TAG}_f -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmct_${TAG}_w -- python3 bench.py $ARGS > gpurun_out/pmct_${TAG}_w.log 2>&1
echo done
