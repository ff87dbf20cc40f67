#!/bin/bash
# Round 4, first GPU call: the parity suite over every kernel path, A/B of the new bit-plane kernels on the roofline leg, kernel stats.
OUT=${1:-gpurun_out/r04a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -x -k "not launcher and not rccl and not two_processes" 2>&1 | tail -80) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for v in "default" "LM_BITS_WAVES=5" "LM_COARSE_BITS=0" "LM_BITPLANES=0"; do
  if [ "$v" = "default" ]; then e=""; else e="$v"; fi
  echo "== $v" >> $OUT/roofline_ab.txt
  env $e timeout 200 python bench.py --roofline-only --no-parity-gate 2>> $OUT/roofline_ab.err | tail -1 >> $OUT/roofline_ab.txt
done
cat $OUT/roofline_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/kernel_stats_roofline_leg.txt | head -12
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 200 > $OUT/bench_steps200.json 2> $OUT/bench_steps200.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
python - <<PY
import json
for f in ("bench_steps200", "bench_driver_flags"):
    try:
        d = json.load(open("$OUT/%s.json" % f))
        print(f, "ms/step %.4f value %.3g" % (d["ms_per_step"], d["value"]), d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], d["roofline"]["other"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 $OUT/bench_steps200.err
