#!/bin/bash
# Round 6: every number DESIGN.md section 7 quotes, in one call on one GPU box.  Usage (GPU box): profiles/r06_final.sh [out_dir] [quick]
OUT=${1:-gpurun_out/r06final}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5) > $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
# (the repeats of the timed region are inside every bench line since round 6: ms_per_step = median of 9, min / max under `repeats`)
timeout 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $OUT/bench_steps200.json 2> $OUT/bench_steps200.err
timeout 300 python bench.py --scaling strong --steps 50 --no-extras --no-cpu-baseline > $OUT/bench_strong_n1_16k.json 2> $OUT/bench_strong_n1_16k.err
if [ -z "$2" ]; then
timeout 300 python bench.py --dry-ranks 8 --steps 8 > $OUT/dry_ranks8.json 2> $OUT/dry_ranks8.err
timeout 200 python profiles/host_profile.py 2>&1 | grep steps > $OUT/host_profile.txt
timeout 200 python profiles/short_run_timeline.py 2>&1 | grep -A1 "^rep" > $OUT/short_run_timeline.txt
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/kernel_stats_roofline_leg.txt > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_bench -o bench -- python $ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --steps 200 > $ROOT/$OUT/bench_under_rocprof.json 2> /dev/null
DB=$(find $ROOT/$OUT/prof_bench -name "*_results.db" | head -1)
[ -n "$DB" ] && python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/kernel_stats_bench_steps200.txt > /dev/null
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT
profiles/pmc_run.sh $OUT/pmc r06 > /dev/null 2>&1
tail -3 $OUT/pytest_gpu.log

python - <<PY
import json
for f in ("bench_default", "bench_driver_flags", "bench_steps200", "bench_strong_n1_16k"):
    try:
        d = json.load(open("$OUT/%s.json" % f))
        print(f, "ms/step %.4f value %.3g" % (d["ms_per_step"], d["value"]), d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], (d["roofline"].get("binding"), d["roofline"].get("useful_valu_frac")))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -4 $OUT/bench_default.err
