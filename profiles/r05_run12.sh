#!/bin/bash
# Round 5, twelfth GPU call: frames in flight of the bench loop (LM_BENCH_DEPTH) at the driver's 20 steps, 50 and 200
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for depth in 16 12 10 8 6; do
  r=""
  for i in 1 2 3 4; do
    v=$(LM_BENCH_DEPTH=$depth python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    r="$r $v"
  done
  v50=$(LM_BENCH_DEPTH=$depth python3 bench.py --steps 50 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  v200=$(LM_BENCH_DEPTH=$depth python3 bench.py --steps 200 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  echo "depth=$depth: steps20 $r | steps50 $v50 | steps200 $v200"
done
