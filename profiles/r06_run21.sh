#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for b in 0; do
  echo "LM_ICP_BUILDS=$b"
  LM_ICP_BUILDS=$b timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-160
  LM_ICP_BUILDS=$b timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done > gpurun_out/r06_run21_builds.txt 2>&1
cat gpurun_out/r06_run21_builds.txt
