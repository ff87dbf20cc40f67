#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl or exchange or shard" > gpurun_out/r06_run18_pytest.log 2>&1
tail -15 gpurun_out/r06_run18_pytest.log
