#!/bin/bash
# Round 5, second GPU call: plan statistics of the bench workload, the refinement with runs of at most 1 / 2 / 3 / 5 members (LM_RUN_MAXK)
OUT=${1:-gpurun_out/r05b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
for k in 5 3 2 1; do
  echo "== LM_RUN_MAXK=$k"; LM_RUN_MAXK=$k timeout 300 python profiles/r05_probe.py > $OUT/probe_k$k.json 2> $OUT/probe_err$k.txt; python - <<PY
import json
d=json.load(open('$OUT/probe_k$k.json'))
for r in d: print(r, d[r][-2:])
PY
done
cd /tmp && export TMPDIR=/tmp
for k in 5 2 1; do
  LM_RUN_MAXK=$k timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$k -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > $ROOT/$OUT/roof$k.json 2> $ROOT/$OUT/err$k.txt
  DB=$(find $ROOT/$OUT/prof$k -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$k.txt > /dev/null
  echo "maxk=$k:"; grep -E "k_local_bits|k_plan_runs" $ROOT/$OUT/stats$k.txt | cut -c1-30,60-150
done
find $ROOT/$OUT -name "*_results.db" -delete
