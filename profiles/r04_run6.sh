#!/bin/bash
# Round 4, sixth GPU call: the defaults (8 frames per launch, one stream, k_dedupe as its own launch) — suite, driver-flag repeats, cfg4.
OUT=${1:-gpurun_out/r04g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -30) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for st in 200 50 20 20 20; do
  timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], d["parity"]["stream"]["steps"][-1], {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect", "host_wait_ms")}, {k: round(v, 3) for k, v in d["stages_ms"].items() if k != "note"})
except Exception as e:
    print("steps $st FAILED", e)
PY
done 2>&1 | tee $OUT/bench_runs.txt
tail -3 $OUT/bench_tmp.err
timeout 600 python profiles/cfg4_full_bank.py > $OUT/cfg4_full_bank.json 2> $OUT/cfg4.err; tail -c 1500 $OUT/cfg4_full_bank.json; tail -3 $OUT/cfg4.err
