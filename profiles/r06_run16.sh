#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r06_icp_trace -o icp -- python $ROOT/profiles/icp_only.py 16 > $ROOT/gpurun_out/r06_icp_trace.log 2>&1
cd $ROOT
python profiles/rocpd_summary.py $(find gpurun_out/r06_icp_trace -name "*_results.db" | head -1) gpurun_out/r06_icp_kernel_stats.txt > /dev/null
grep -i "icp" gpurun_out/r06_icp_kernel_stats.txt | cut -c1-200
find gpurun_out/r06_icp_trace -name "*_results.db" -delete
grep device_ms gpurun_out/r06_icp_trace.log | cut -c1-200
