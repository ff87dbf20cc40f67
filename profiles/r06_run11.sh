#!/bin/bash
# round 6: kernel trace of the icp leg (bench.icp_bench: 16 hypotheses) with the team kernel, and of the pipeline leg
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r06_icp_trace -o icp -- python $ROOT/profiles/icp_only.py 16 > $ROOT/gpurun_out/r06_icp_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r06_pipe_trace -o pipe -- python $ROOT/profiles/pipeline_only.py 5 > $ROOT/gpurun_out/r06_pipe_trace.log 2>&1
cd $ROOT
tail -3 gpurun_out/r06_icp_trace.log; tail -3 gpurun_out/r06_pipe_trace.log
find gpurun_out/r06_icp_trace gpurun_out/r06_pipe_trace -name "*kernel_stats.csv" | head
for f in $(find gpurun_out/r06_icp_trace gpurun_out/r06_pipe_trace -name "*kernel_stats.csv"); do echo $f; head -16 $f | cut -c1-200; done
