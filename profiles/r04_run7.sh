#!/bin/bash
# Round 4, seventh GPU call: launch rule at the start of a stream (idle GPU = everything launched has finished), coarse pass with 16 loads in flight.
OUT=${1:-gpurun_out/r04h}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined or fixture or config1 or sharded" 2>&1 | tail -8) > $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 20 20 20; do
    env "${envs[@]}" timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline "$@" > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$label steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect", "host_wait_ms")}, {k: round(v, 3) for k, v in d["stages_ms"].items() if k != "note"})
except Exception as e:
    print("$label steps $st FAILED", e)
PY
  done
}
{
run default X=1 --
run deep LM_CBITS_DEEP=1 --
run batch4 X=1 -- --batch 4
} 2>&1 | tee $OUT/bench_ab.txt
timeout 200 python profiles/short_run_timeline.py 2>&1 | grep -A1 "^rep" > $OUT/short_run_timeline.txt; cut -c1-1200 $OUT/short_run_timeline.txt
for v in "default" "LM_CBITS_DEEP=1"; do
  if [ "$v" = "default" ]; then e="X=1"; else e="$v"; fi
  echo "== $v" >> $OUT/roofline_ab.txt
  env $e timeout 200 python bench.py --roofline-only --no-parity-gate 2>> $OUT/roofline_ab.err | tail -1 >> $OUT/roofline_ab.txt
done
cat $OUT/roofline_ab.txt
