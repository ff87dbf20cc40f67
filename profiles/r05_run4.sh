#!/bin/bash
# Round 5, fourth GPU call: horizontal pairs with interleaved lanes (LM_LOCAL_RUNS=2) — parity subset, kernel times, PMC
OUT=${1:-gpurun_out/r05d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
LM_LOCAL_RUNS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture_banks or planted or edge_cases or feature_count" > $OUT/pytest_subset.log 2>&1; echo "pytest subset (pairs) rc=$?"; tail -3 $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
for blocks in 0 1024; do
 for v in 2 0; do
  LM_LOCAL_BLOCKS=$blocks LM_LOCAL_RUNS=$v timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof${v}_$blocks -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > /dev/null 2> $ROOT/$OUT/err.txt
  DB=$(find $ROOT/$OUT/prof${v}_$blocks -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats${v}_$blocks.txt > /dev/null
  echo "blocks=$blocks runs=$v:"; grep -E 'k_local_bits|k_plan' $ROOT/$OUT/stats${v}_$blocks.txt | cut -c1-30,60-150
 done
done
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
LM_LOCAL_RUNS=2 bash profiles/pmc_run.sh $OUT/pmc2 r05pairs k_local_bits,k_plan_pairs > /dev/null 2>&1
cat $OUT/pmc2/pmc_r05pairs.txt | head -30
