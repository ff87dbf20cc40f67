#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
(timeout 900 python -m pytest tests -m gpu -q -x -k "icp or pose or refine or pipeline" 2>&1 | tail -3)
for r in 1 2; do timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-120; done
timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
TEAM_MEMBERS=0 timeout 300 python profiles/r06_icp_team.py 16 2>&1 | grep "k_icp_knn" | head -2
