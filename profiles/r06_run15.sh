#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_run15_pytest.log 2>&1
tail -5 gpurun_out/r06_run15_pytest.log
