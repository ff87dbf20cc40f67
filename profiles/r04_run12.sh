#!/bin/bash
# Round 4, twelfth GPU call: the pair stream of the top level from pixel tiles (top_bits_tile_body): parity, kernel times (k_fe_bits split), stream.
OUT=${1:-gpurun_out/r04p}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -k "frontend or masks or bit_planes or fixture or config1 or reference_lines or stream or edge_cases or refinement_paths or planted or boundaries" 2>&1 | tail -12) > $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
for st in 200 20 20 20; do
  timeout 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_tmp.json")); print("steps $st: ms/step %.4f" % d["ms_per_step"], d["config"].get("frames_per_launch_mean_timed"), d["parity_checked"], {k: round(v, 3) for k, v in d["stages_ms"].items() if k != "note"})
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
