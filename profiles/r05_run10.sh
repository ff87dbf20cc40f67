#!/bin/bash
# Round 5, tenth GPU call: early partial batches per burst of a tight stream (LM_EARLY_BATCHES x LM_FIRST_BATCH) at the driver's 20 steps, 50 and 200
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for cfg in "1 3" "2 3" "3 3" "2 2" "3 2" "2 4" "1 3"; do
  set -- $cfg
  r=""
  for i in 1 2 3 4; do
    v=$(LM_EARLY_BATCHES=$1 LM_FIRST_BATCH=$2 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    r="$r $v"
  done
  v50=$(LM_EARLY_BATCHES=$1 LM_FIRST_BATCH=$2 python3 bench.py --steps 50 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  v200=$(LM_EARLY_BATCHES=$1 LM_FIRST_BATCH=$2 python3 bench.py --steps 200 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  echo "early=$1 first=$2: steps20 $r | steps50 $v50 | steps200 $v200"
done
