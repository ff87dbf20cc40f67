#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TEAM_ROWS=1 LM_ICP_TEAM=16 timeout 300 python profiles/r06_icp_team.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_run5_icp_team.txt
cut -c1-420 gpurun_out/r06_run5_icp_team.txt
