#!/bin/bash
# Round 4, third GPU call: helper threads of the streamed path (sliced staging copy, result lists of a batch's later frames), one upload per frame.
OUT=${1:-gpurun_out/r04c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -k "not launcher and not rccl and not two_processes" 2>&1 | tail -60) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 200 python profiles/host_profile.py 2>&1 | grep steps > $OUT/host_profile.txt; cat $OUT/host_profile.txt
timeout 200 python profiles/short_run_timeline.py 2>&1 | grep -A1 "^rep" > $OUT/short_run_timeline.txt; cut -c1-900 $OUT/short_run_timeline.txt
for v in "default" "LM_HOST_THREADS=0" "LM_HOST_THREADS=5" "LM_ASYNC_COLLECT=0"; do
  if [ "$v" = "default" ]; then e=""; else e="$v"; fi
  for st in 200 20 20; do
    env $e timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$v steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 4) for k, v in d["host_wall_ms"].items()})
except Exception as e:
    print("$v steps $st FAILED", e)
PY
  done
done 2>&1 | tee $OUT/bench_ab.txt
tail -3 $OUT/bench_tmp.err
cd /tmp && export TMPDIR=/tmp
LM_FE_BITS_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python - <<PY
import sqlite3
con = sqlite3.connect("$DB")
rows = con.execute("select name, grid_x, count(*), avg(duration), min(duration) from kernels where name like '%k_fe_%' group by name, grid_x order by name, grid_x").fetchall()
for r in rows: print("%-40s grid %8d calls %4d avg %.2f us min %.2f us" % (r[0][:40], r[1], r[2], r[3] / 1e3, r[4] / 1e3))
PY
find $ROOT/$OUT -name "*_results.db" -delete
