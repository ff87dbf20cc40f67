#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lat profiles/r06_latency_microbench.hip && /tmp/lat > gpurun_out/r06_latency_microbench.txt 2>&1
cat gpurun_out/r06_latency_microbench.txt
