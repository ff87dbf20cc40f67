#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r06_pipe_trace -o pipe -- python $ROOT/profiles/pipeline_only.py 5 > $ROOT/gpurun_out/r06_pipe_trace.log 2>&1
cd $ROOT
python profiles/rocpd_summary.py $(find gpurun_out/r06_pipe_trace -name "*_results.db" | head -1) gpurun_out/r06_pipe_kernel_stats.txt > /dev/null
grep -i "icp" gpurun_out/r06_pipe_kernel_stats.txt | cut -c1-200
find gpurun_out/r06_pipe_trace -name "*_results.db" -delete
