#!/bin/bash
# Round 4, fourteenth GPU call: the coarse pass with two frames of a batch per wave (k_coarse_bits2; LM_COARSE_PAIRS=0 = one frame per wave): parity + A/B.
OUT=${1:-gpurun_out/r04x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined or fixture or config1 or config3 or config4 or edge_cases or refinement_paths or planted or boundaries or sharded or pipeline_equals" 2>&1 | tail -12) > $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 20 20 20; do
    env "${envs[@]}" timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc "$@" > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("$label steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 3) for k, v in d["stages_ms"].items() if k != "note"})
except Exception as e:
    print("$label steps $st FAILED", e)
PY
  done
}
{
run pairs X=1 --
run one_frame_per_wave LM_COARSE_PAIRS=0 --
} 2>&1 | tee $OUT/bench_ab.txt
