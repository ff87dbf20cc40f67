#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pose_refine or icp or pipeline" > gpurun_out/r06_run8_pytest.log 2>&1
tail -5 gpurun_out/r06_run8_pytest.log
TEAM_ROWS=1 LM_ICP_TEAM=16 timeout 300 python profiles/r06_icp_team.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_run8_icp_team.txt
cut -c1-420 gpurun_out/r06_run8_icp_team.txt
