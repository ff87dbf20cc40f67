#!/bin/bash
# Round 5, ninth GPU call: horizontal pairs on the same lanes (k_plan_pairs + pair items of k_local_bits): parity, then A/B against LM_LOCAL_PAIRS=0
OUT=${1:-gpurun_out/r05i}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "horizontal_pairs or fixture_banks or planted or edge_cases or feature_count or ceiling" > $OUT/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  LM_LOCAL_PAIRS=$v timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$v -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > $ROOT/$OUT/roof$v.json 2> $ROOT/$OUT/err$v.txt
  DB=$(find $ROOT/$OUT/prof$v -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$v.txt > /dev/null
  echo "pairs=$v:"; grep -E "k_local_bits|k_plan_pairs|k_coarse_bits|k_dedupe" $ROOT/$OUT/stats$v.txt | cut -c1-30,60-150
done
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
LM_LOCAL_PAIRS=1 bash profiles/pmc_run.sh $OUT/pmc r05pairs k_local_bits,k_plan_pairs > /dev/null 2>&1
head -28 $OUT/pmc/pmc_r05pairs.txt
