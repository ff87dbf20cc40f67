#!/bin/bash
# Round 5, fourteenth GPU call: does the GPU's performance level (rocm-smi --setperflevel) matter for the short stream?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -20
run() {
  r=""
  for i in 1 2 3 4; do
    v=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    r="$r $v"
  done
  v200=$(python3 bench.py --steps 200 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  echo "$1: steps20 $r | steps200 $v200"
}
run auto
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel --showclocks 2>&1 | grep -i "perf\|sclk\|mclk" | head -8
run high
rocm-smi --setperflevel auto 2>&1 | tail -2
