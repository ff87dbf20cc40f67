#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python profiles/r06_icp_solo.py 0 > gpurun_out/r06_run2_icp_solo.txt 2>&1
cat gpurun_out/r06_run2_icp_solo.txt | cut -c1-300
