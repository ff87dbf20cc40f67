#!/bin/bash
# Round 4, eighteenth GPU call: k_fe_stage at 6 / 7 / 8 waves per SIMD (= persistent workgroups per CU): builds with -DLM_FE_WAVES=7 / 8 beside the default.
OUT=${1:-gpurun_out/r04fw}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for w in 6 7 8; do
  LIB=$ROOT/6dpose_amd/libamdlinemod.so; [ $w != 6 ] && LIB=$ROOT/6dpose_amd/libamdlinemod_w$w.so
  AMD_LINEMOD_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$w -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > /dev/null 2> $ROOT/$OUT/err.txt
  DB=$(find $ROOT/$OUT/prof$w -name "*_results.db" | head -1)
  python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats$w.txt > /dev/null
  echo "waves $w: $(grep k_fe_stage $ROOT/$OUT/stats$w.txt | cut -c60-140)"
done
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
for w in 7; do
  AMD_LINEMOD_LIB=$ROOT/6dpose_amd/libamdlinemod_w$w.so timeout 300 python bench.py --steps 200 --warmup 5 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $w 200 steps: %.4f' % d['ms_per_step'], d['parity_checked'], d['stages_ms']['frontend_ms'])"
done
timeout 300 python bench.py --steps 200 --warmup 5 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves 6 200 steps: %.4f' % d['ms_per_step'], d['parity_checked'], d['stages_ms']['frontend_ms'])"
