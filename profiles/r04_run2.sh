#!/bin/bash
# Round 4, second GPU call: the front end writing bit planes directly (tests of every path again), where the host thread's time goes, PMC passes.
OUT=${1:-gpurun_out/r04b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -k "not launcher and not rccl and not two_processes" 2>&1 | tail -60) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for v in "default" "LM_FE_BITS=0"; do
  if [ "$v" = "default" ]; then e=""; else e="$v"; fi
  echo "== $v" >> $OUT/roofline_ab.txt
  env $e timeout 200 python bench.py --roofline-only --no-parity-gate 2>> $OUT/roofline_ab.err | tail -1 >> $OUT/roofline_ab.txt
done
cat $OUT/roofline_ab.txt
timeout 200 python profiles/host_profile.py 2>&1 | grep steps > $OUT/host_profile.txt; cat $OUT/host_profile.txt
timeout 200 python profiles/short_run_timeline.py 2>&1 | grep -A1 "^rep" > $OUT/short_run_timeline.txt; cat $OUT/short_run_timeline.txt
for v in "default" "LM_ASYNC_COLLECT=1"; do
  if [ "$v" = "default" ]; then e=""; else e="$v"; fi
  for st in 200 20; do
    env $e timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$v steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["host_wall_ms"])
except Exception as e:
    print("$v steps $st FAILED", e)
PY
  done
done 2>&1 | tee $OUT/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/kernel_stats_roofline_leg.txt | head -12
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
profiles/pmc_run.sh $OUT/pmc r04 > /dev/null 2>&1
cat $OUT/pmc/passes.txt; head -70 $OUT/pmc/pmc_r04.txt
