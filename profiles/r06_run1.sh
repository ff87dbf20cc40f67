#!/bin/bash
# round 6, first GPU call: k_icp_solo correctness (the ICP / pipeline GPU tests) and the icp leg per LM_ICP_SOLO setting
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pose_refine or icp or pipeline" > gpurun_out/r06_run1_pytest.log 2>&1
tail -5 gpurun_out/r06_run1_pytest.log
timeout 600 python profiles/r06_icp_solo.py > gpurun_out/r06_run1_icp_solo.txt 2>&1
cat gpurun_out/r06_run1_icp_solo.txt | cut -c1-400
