#!/bin/bash
# PMC counter passes for the two dominant kernels (each pass is its own rocprofv3 run with --kernel-trace only).
# usage (on the GPU box): bash tools/pmc_run.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o p$i --output-format csv -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
ls -R $OUT | head -40
