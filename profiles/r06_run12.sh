#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for sl in 1 0; do LM_ICP_SLICED=$sl timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1; done > gpurun_out/r06_run12_pipeline.txt
cat gpurun_out/r06_run12_pipeline.txt
