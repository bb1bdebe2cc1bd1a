#!/bin/bash
# usage: tools/pmc_conv.sh TAG "--only C64-K128-3x3-H56 --kinds fprop"   (run on the GPU box; writes gpurun_out/pmc_TAG_*)
TAG=$1; shift
ARGS="$*"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $REPO/gpurun_out/pmc_${TAG}_$name -o p --output-format csv -- python $REPO/tools/conv_bench.py $ARGS --iters 2 > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
cd $REPO
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_sq1 gpurun_out/pmc_${TAG}_sq2 gpurun_out/pmc_${TAG}_sq3 gpurun_out/pmc_${TAG}_tcc --match igemm > gpurun_out/pmc_${TAG}_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_sq1 gpurun_out/pmc_${TAG}_sq2 gpurun_out/pmc_${TAG}_sq3 gpurun_out/pmc_${TAG}_tcc --match wgrad >> gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat gpurun_out/pmc_${TAG}_summary.txt
