#!/bin/bash
# Round 4, fifth GPU call: duplicate removal inside k_local_bits, batch sizes on the single-stream layout.
OUT=${1:-gpurun_out/r04e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -k "not launcher and not rccl and not two_processes" 2>&1 | tail -60) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
run() {  # label, env..., -- bench args
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for st in 200 20 20; do
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
run no_inline LM_INLINE_DEDUPE=0 --
run batch8 X=1 -- --batch 8
run batch8_serial1 LM_SERIAL=1 -- --batch 8
run batch8_noinline LM_INLINE_DEDUPE=0 -- --batch 8
} 2>&1 | tee $OUT/bench_ab.txt
tail -3 $OUT/bench_tmp.err
for v in "default" "LM_INLINE_DEDUPE=0"; do
  if [ "$v" = "default" ]; then e="X=1"; else e="$v"; fi
  echo "== $v" >> $OUT/roofline_ab.txt
  env $e timeout 200 python bench.py --roofline-only --no-parity-gate 2>> $OUT/roofline_ab.err | tail -1 >> $OUT/roofline_ab.txt
done
cat $OUT/roofline_ab.txt
timeout 300 python bench.py --dry-ranks 8 --steps 8 > $OUT/dry_ranks8.json 2> $OUT/dry_ranks8.err; cut -c1-600 $OUT/dry_ranks8.json; tail -3 $OUT/dry_ranks8.err
