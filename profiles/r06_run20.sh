#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pose_refine or icp or pipeline" > gpurun_out/r06_run20_pytest.log 2>&1
tail -3 gpurun_out/r06_run20_pytest.log
bash profiles/r06_run16.sh
