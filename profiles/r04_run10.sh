#!/bin/bash
# Round 4, tenth GPU call: bit-sliced 3x3 vote and 5x5 median in the front end (k_fe_stage): parity tests of the front end, kernel times, stream.
OUT=${1:-gpurun_out/r04m}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -k "frontend or masks or bit_planes or add_template or fixture or config1 or reference_lines or stream or edge_cases" 2>&1 | tail -8) > $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
for st in 200 20 20 20; do
  timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 3) for k, v in d["stages_ms"].items() if k != "note"})
except Exception as e:
    print("steps $st FAILED", e)
PY
done 2>&1 | tee $OUT/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/kernel_stats_roofline_leg.txt > /dev/null
find $ROOT/$OUT -name "*_results.db" -delete
head -8 $ROOT/$OUT/kernel_stats_roofline_leg.txt | cut -c1-150
