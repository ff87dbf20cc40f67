#!/bin/bash
# Round 5, seventh GPU call: whole GPU suite + default bench line (extras in the headline's loop, stream gate at the timed region's launch rule)
OUT=${1:-gpurun_out/r05g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time | head -1; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline kernel_ms', d['roofline'].get('kernel_ms'))
print('stream gate region', json.dumps(d['parity']['stream'].get('timed_region_rule'))[:600])
e=d['extras']
for k in ('one_candidate_per_template','real_fixture','strong_scaling_reference','strong_scaling_proxy'):
    print(k, json.dumps(e.get(k))[:700])
print('icp', e['icp']['device_ms'], 'pipeline', e['pipeline']['total_ms'], e['pipeline']['icp_ms'])
PY
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench20.json 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench20.json').read().strip().splitlines()[-1]); print('driver flags: ms_per_step', d['ms_per_step'])"
