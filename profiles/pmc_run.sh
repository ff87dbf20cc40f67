#!/bin/bash
# Counter passes for the hot kernels (run on the GPU box through gpurun).  One rocprofv3 run per
# --pmc group (SQ 8 slots, TCC 4 slots; FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), kernel-trace only,
# as MI355X_MICROARCH.md §rocprofv3 PMC slots prescribes.  Usage: pmc_run.sh <out_dir> <tag> [kernels]
# The profiled command is bench.py's roofline leg alone (--roofline-only): every k_local / k_coarse / k_fe_stage / k_dedupe dispatch
# of the run but the set-up probe is one of the launches bench.py's `roofline` object describes (frames_per_launch frames each).
set -u
OUT=${1:-gpurun_out/pmc}; TAG=${2:-r03}; KERNELS=${3:-k_local_bits,k_coarse_bits,k_fe_stage,k_fe_bits,k_dedupe,k_local,k_coarse,k_pack_bits,k_pack_top}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $ROOT/bench.py --roofline-only --no-parity-gate"}
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/$OUT/p$i -o $TAG -- $CMD > $ROOT/$OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $grp" >> $ROOT/$OUT/passes.txt
done
python - <<PY
import sys
sys.path.insert(0, "$ROOT/profiles")
import rocpd_pmc
rocpd_pmc.main("$ROOT/$OUT/p*/*_results.db", "$ROOT/$OUT/pmc_$TAG.txt", tuple("$KERNELS".split(",")))
PY
find $ROOT/$OUT -name "*_results.db" -delete
