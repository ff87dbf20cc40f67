#!/bin/bash
# round 6: k_icp_team correctness (the ICP / pipeline GPU tests) and the icp leg per team size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pose_refine or icp or pipeline" > gpurun_out/r06_run4_pytest.log 2>&1
tail -5 gpurun_out/r06_run4_pytest.log
for t in 16 8; do
  LM_ICP_TEAM=$t timeout 300 python profiles/r06_icp_team.py 16 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06_run4_icp_team.txt
cut -c1-420 gpurun_out/r06_run4_icp_team.txt
