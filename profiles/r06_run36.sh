#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 0 16 20 24 32; do
  echo "LM_KNN_BLOCKS=$b"
  for r in 1 2 3; do LM_KNN_BLOCKS=$b timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-90; done
  LM_KNN_BLOCKS=$b timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
done
