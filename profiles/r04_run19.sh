#!/bin/bash
# Round 4, nineteenth GPU call: upper bound of pair sharing in k_local_bits — a build (-DLM_DIAG_SKIP_PAIR_LOADS, WRONG results) whose odd groups skip 31 % of the loads.
OUT=${1:-gpurun_out/r04skip}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for v in default skip; do
  LIB=$ROOT/6dpose_amd/libamdlinemod.so; [ $v = skip ] && LIB=$ROOT/6dpose_amd/libamdlinemod_skip.so
  AMD_LINEMOD_LIB=$LIB timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$v -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > /dev/null 2> $ROOT/$OUT/err.txt
  DB=$(find $ROOT/$OUT/prof$v -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$v.txt > /dev/null
  echo "$v: $(grep k_local_bits $ROOT/$OUT/stats$v.txt | cut -c60-140)"
done
find $ROOT/$OUT -name "*_results.db" -delete
