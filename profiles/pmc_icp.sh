#!/bin/bash
# Counter passes for the ICP kernels (run on the GPU box through gpurun); one rocprofv3 run per --pmc group,
# kernel-trace only (MI355X_MICROARCH.md §rocprofv3 PMC slots).  Usage: pmc_icp.sh <out_dir>
set -u
OUT=${1:-gpurun_out/pmc_icp}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
CMD=${LM_PMC_CMD:-"python $ROOT/profiles/pipeline_only.py 3"}
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/$OUT/p$i -o icp -- $CMD > $ROOT/$OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $grp" >> $ROOT/$OUT/passes.txt
done
