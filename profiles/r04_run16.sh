#!/bin/bash
# Round 4, sixteenth GPU call: the duplicate removal of a batch on a side queue (LM_SERIAL=3) against everything on one queue (LM_SERIAL=2, default) — A/B on one box.
OUT=${1:-gpurun_out/r04s3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(LM_SERIAL=3 timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined or config1 or sharded or exchange" 2>&1 | tail -4) > $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 200 20 20 20; do
    env "${envs[@]}" timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc "$@" > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$label steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect")})
except Exception as e:
    print("$label steps $st FAILED", e)
PY
  done
}
{
run dedupe_aside LM_SERIAL=3 --
run one_queue LM_SERIAL=2 --
run dedupe_aside_again LM_SERIAL=3 --
} 2>&1 | tee $OUT/bench_ab.txt
