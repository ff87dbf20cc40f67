#!/bin/bash
# Round 5, thirteenth GPU call: the upload of a streamed frame on two copy queues (LM_COPY_STREAMS=2, the new default) against one
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "live_stream or helper_threads or stream or batch" 2>&1 | tail -3
for cs in 2 1 2 1; do
  r=""
  for i in 1 2 3 4; do
    v=$(LM_COPY_STREAMS=$cs python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    r="$r $v"
  done
  v50=$(LM_COPY_STREAMS=$cs python3 bench.py --steps 50 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f h2d %.4f' % (d['ms_per_step'], d['stages_ms']['h2d_ms']))")
  v200=$(LM_COPY_STREAMS=$cs python3 bench.py --steps 200 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  echo "copy streams=$cs: steps20 $r | steps50 $v50 | steps200 $v200"
done
