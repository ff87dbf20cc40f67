#!/bin/bash
# Round 4, eighth GPU call: stage timing events without the system-scope fence and one record between two kernels (LM_STAGE_EVENTS=2 = round 3's records) — A/B on one box.
OUT=${1:-gpurun_out/r04k}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined or fixture or config1 or timings" 2>&1 | tail -8) > $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 20 20 20; do
    env "${envs[@]}" timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc "$@" > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$label steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect", "host_wait_ms")}, {k: round(v, 3) for k, v in d["stages_ms"].items() if k != "note"}, "roofline kernel_ms %.4f" % d["roofline"]["kernel_ms"])
except Exception as e:
    print("$label steps $st FAILED", e)
PY
  done
}
{
run lean X=1 --
run round3_events LM_STAGE_EVENTS=2 --
run lean_again X=1 --
} 2>&1 | tee $OUT/bench_ab.txt
timeout 200 python profiles/short_run_timeline.py 2>&1 | grep -A1 "^rep" > $OUT/short_run_timeline.txt; cut -c1-900 $OUT/short_run_timeline.txt
bash $ROOT/profiles/r04_trace20.sh $OUT/trace 2>&1 | tail -48
find $ROOT/$OUT -name "*_results.db" -delete
