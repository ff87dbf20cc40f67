#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_run34
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -k "icp or pose or refine or pipeline" 2>&1 | tail -3)
for r in 1 2; do timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-120; done
timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_icp -o icp -- python $GRAFT_REPO_ROOT/profiles/icp_only.py 16 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_pipe -o pipe -- python $GRAFT_REPO_ROOT/profiles/pipeline_only.py 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/rocpd_summary.py $(find $OUT/trace_icp -name "*_results.db" | head -1) $OUT/kernel_stats_icp_leg.txt > /dev/null
python profiles/rocpd_summary.py $(find $OUT/trace_pipe -name "*_results.db" | head -1) $OUT/kernel_stats_pipeline_leg.txt > /dev/null
rm -rf $OUT/trace_icp $OUT/trace_pipe
grep "knn" $OUT/kernel_stats_icp_leg.txt | cut -c1-150
grep "knn" $OUT/kernel_stats_pipeline_leg.txt | cut -c1-150
