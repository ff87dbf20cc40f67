#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 0 1 2 3; do
  echo "LM_ICP_BUILDS=$b"
  for r in 1 2; do LM_ICP_BUILDS=$b timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-120; done
  LM_ICP_BUILDS=$b timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
done
