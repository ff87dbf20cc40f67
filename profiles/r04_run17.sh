#!/bin/bash
OUT=${1:-gpurun_out/r04dd}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for nbk in 0 64 128 32; do
LM_DEDUPE_BLOCKS=$nbk timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$nbk -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > /dev/null 2> $ROOT/$OUT/err.txt
DB=$(find $ROOT/$OUT/prof$nbk -name "*_results.db" | head -1)
python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$nbk.txt > /dev/null
echo "LM_DEDUPE_BLOCKS=$nbk: $(grep k_dedupe $ROOT/$OUT/stats$nbk.txt | cut -c60-130)"
done
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
LM_DEDUPE_BLOCKS=64 timeout 600 python -m pytest tests -m gpu -q -k "stream or planted or edge_cases" 2>&1 | tail -2
