#!/bin/bash
# Round 5, third GPU call: PMC of k_local_bits with runs (LM_RUN_MAXK=5) and with every candidate a single through the same lists (=1); grid sizes
OUT=${1:-gpurun_out/r05c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
for k in 5 1; do
  LM_RUN_MAXK=$k bash profiles/pmc_run.sh $OUT/pmc$k r05k$k k_local_bits,k_plan_runs > /dev/null 2>&1
  echo "== maxk=$k"; cat $OUT/pmc$k/pmc_r05k$k.txt | head -60
done
cd /tmp && export TMPDIR=/tmp
for blocks in 512 1024 2048; do
 for k in 5 1; do
  LM_LOCAL_BLOCKS=$blocks LM_RUN_MAXK=$k timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof${k}_$blocks -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > /dev/null 2> $ROOT/$OUT/err.txt
  DB=$(find $ROOT/$OUT/prof${k}_$blocks -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats${k}_$blocks.txt > /dev/null
  echo "blocks=$blocks maxk=$k: $(grep -E 'k_local_bits' $ROOT/$OUT/stats${k}_$blocks.txt | cut -c60-150)"
 done
done
find $ROOT/$OUT -name "*_results.db" -delete
