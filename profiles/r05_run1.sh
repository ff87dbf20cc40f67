#!/bin/bash
# Round 5, first GPU call: vertical runs in the bit-plane refinement (k_plan_runs + run items of k_local_bits) — parity, then A/B against LM_LOCAL_RUNS=0.
OUT=${1:-gpurun_out/r05a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vertical_runs or fixture_banks or planted or edge_cases or feature_count" > $OUT/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  LM_LOCAL_RUNS=$v timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$v -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > $ROOT/$OUT/roof$v.json 2> $ROOT/$OUT/err$v.txt
  DB=$(find $ROOT/$OUT/prof$v -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$v.txt > /dev/null
  echo "runs=$v:"; grep -E "k_local_bits|k_plan_runs|k_coarse_bits|k_dedupe|k_fe" $ROOT/$OUT/stats$v.txt | cut -c1-30,60-150
done
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
for v in 1 0; do
  for steps in 200 20; do
    LM_LOCAL_RUNS=$v timeout 300 python bench.py --steps $steps --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_runs${v}_steps$steps.json 2>> $OUT/bench_err.txt
    python -c "
import json,sys
d=json.loads(open('$OUT/bench_runs${v}_steps$steps.json').read().strip().splitlines()[-1])
print('runs=$v steps=$steps ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline'].get('frac'))"
  done
done
