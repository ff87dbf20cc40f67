#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python profiles/cfg4_full_bank.py 2> gpurun_out/r06_cfg4.err | tail -1 > gpurun_out/r06_cfg4_full_bank.json
tail -3 gpurun_out/r06_cfg4.err; cut -c1-1500 gpurun_out/r06_cfg4_full_bank.json
