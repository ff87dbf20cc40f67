#!/bin/bash
# Round 4, fifteenth GPU call: staging copy of a streamed frame with non-temporal stores (LM_NT_COPY=0: memcpy) — A/B on one box.
OUT=${1:-gpurun_out/r04nt}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined" 2>&1 | tail -4) > $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 20 20 20 20; do
    env "${envs[@]}" timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate "$@" > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$label steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect")})
except Exception as e:
    print("$label steps $st FAILED", e)
PY
  done
}
{
run nt_stores X=1 --
run memcpy LM_NT_COPY=0 --
run nt_stores_again X=1 --
} 2>&1 | tee $OUT/bench_ab.txt
timeout 200 python profiles/host_profile.py 2>&1 | grep steps | cut -c1-330
LM_NT_COPY=0 timeout 200 python profiles/host_profile.py 2>&1 | grep steps | cut -c1-330
