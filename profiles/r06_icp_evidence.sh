#!/bin/bash
# Round 6 evidence for the ICP rows: kernel trace (rocprofv3 --kernel-trace --stats) of the icp leg (16 hypotheses, profiles/icp_only.py) and of
# the pipeline leg (profiles/pipeline_only.py), PMC passes (separate runs, kernel-trace only) of the icp leg for k_icp_team / k_icp_knn, the
# per-member phase split of the team kernel, and the same leg with one launch per evaluation (LM_ICP_SLICED=1: rounds 1-5) for comparison.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_icp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_icp -o icp -- python $ROOT/profiles/icp_only.py 16 > $OUT/icp_leg.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_pipe -o pipe -- python $ROOT/profiles/pipeline_only.py 5 > $OUT/pipeline_leg.json 2> /dev/null
LM_ICP_SLICED=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_icp_sliced -o icp -- python $ROOT/profiles/icp_only.py 16 > $OUT/icp_leg_sliced.json 2> /dev/null
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC" ; do
  # (no memory-side pass: with GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum rocprofv3 aborted on this command — signal 6 after the
  # run, killed by the timeout; the ICP kernels work out of LDS and registers, their HBM traffic is the clouds once)
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc$i -o icp -- python $ROOT/profiles/icp_only.py 16 > $OUT/pmc$i.log 2>&1
  echo "pass $i rc=$? : $grp" >> $OUT/pmc_passes.txt
done
cd $ROOT
python profiles/rocpd_summary.py $(find $OUT/trace_icp -name "*_results.db" | head -1) $OUT/kernel_stats_icp_leg.txt > /dev/null
python profiles/rocpd_summary.py $(find $OUT/trace_pipe -name "*_results.db" | head -1) $OUT/kernel_stats_pipeline_leg.txt > /dev/null
python profiles/rocpd_summary.py $(find $OUT/trace_icp_sliced -name "*_results.db" | head -1) $OUT/kernel_stats_icp_leg_sliced.txt > /dev/null
python - <<PY
import sys
sys.path.insert(0, "profiles")
import rocpd_pmc
import glob, shutil, os
dbs = glob.glob("$OUT/pmc*/**/*_results.db", recursive=True)
os.makedirs("$OUT/pmc_dbs", exist_ok=True)
for n, f in enumerate(dbs): shutil.copy(f, "$OUT/pmc_dbs/%d_results.db" % n)
rocpd_pmc.main("$OUT/pmc_dbs/*_results.db", "$OUT/pmc_icp.txt", kernels=("k_icp_team", "k_icp_knn", "k_icp_voxel_wide", "k_icp_grid_wide", "k_icp_points_fused", "k_icp_normals"))
PY
TEAM_MEMBERS=1 TEAM_ROWS=1 timeout 300 python profiles/r06_icp_team.py 16 2>&1 | grep -v amdgpu.ids > $OUT/team_phases.txt
find $OUT -name "*_results.db" -delete; rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc_dbs $OUT/trace_icp $OUT/trace_pipe $OUT/trace_icp_sliced
ls $OUT; head -14 $OUT/kernel_stats_icp_leg.txt | cut -c1-160; head -12 $OUT/pmc_icp.txt
