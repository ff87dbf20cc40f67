#!/bin/bash
# Round 5, eleventh GPU call: k_coarse_bits with sixteen loads in flight at 4 / 5 waves per SIMD against eight at 6 (default) and 7; 2k and 16k banks
OUT=${1:-gpurun_out/r05k}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for w in 6 4 5 7; do
  LM_COARSE_WAVES=$w timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$w -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > /dev/null 2> $ROOT/$OUT/err.txt
  DB=$(find $ROOT/$OUT/prof$w -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$w.txt > /dev/null
  echo "2k waves=$w: $(grep -E 'k_coarse_bits' $ROOT/$OUT/stats$w.txt | cut -c60-150)"
done
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
for w in 6 4 5; do
  LM_COARSE_WAVES=$w python3 bench.py --scaling strong --steps 30 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16k waves=$w: ms_per_step', d['ms_per_step'], d['stages_ms']['coarse_ms'], d['stages_ms']['local_ms'])"
done
