#!/bin/bash
# Round 4, ninth GPU call: helper threads of the staging copy (LM_HOST_THREADS 3 / 5 / 7) and the size of the first batch of a tight stream (LM_FIRST_BATCH) — A/B on one box.
OUT=${1:-gpurun_out/r04l}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined" 2>&1 | tail -8) > $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 20 20 20 20; do
    env "${envs[@]}" timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate "$@" > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$label steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect", "host_wait_ms")})
except Exception as e:
    print("$label steps $st FAILED", e)
PY
  done
}
{
run default X=1 --
run threads5 LM_HOST_THREADS=5 --
run threads7 LM_HOST_THREADS=7 --
run first3 LM_FIRST_BATCH=3 --
run first4_threads7 LM_FIRST_BATCH=4 LM_HOST_THREADS=7 --
run default_again X=1 --
} 2>&1 | tee $OUT/bench_ab.txt
