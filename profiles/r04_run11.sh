#!/bin/bash
# Round 4, eleventh GPU call: one early batch per burst (20-step series of 8 for the spread), k_fe_bits split into its two jobs under the profiler.
OUT=${1:-gpurun_out/r04n}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -k "stream or pipelined" 2>&1 | tail -8) > $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
for st in 200 20 20 20 20 20 20 20 20; do
  timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), {k: round(v, 4) for k, v in d["host_wall_ms"].items() if k in ("submit", "collect")})
except Exception as e:
    print("steps $st FAILED", e)
PY
done 2>&1 | tee $OUT/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
LM_FE_BITS_SPLIT=1 timeout 300 rocprofv3 --kernel-trace -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
python - <<PY
import sqlite3
con = sqlite3.connect("$DB")
rows = con.execute("select start, end, name from kernels order by start").fetchall()
fe = [(e - s) / 1e3 for s, e, n in rows if "k_fe_bits" in n]
st = [(e - s) / 1e3 for s, e, n in rows if "k_fe_stage" in n]
print("k_fe_bits launches (us), in order:", [round(x, 1) for x in fe[-12:]])
print("k_fe_stage launches (us), in order:", [round(x, 1) for x in st[-12:]])
PY
find $ROOT/$OUT -name "*_results.db" -delete
