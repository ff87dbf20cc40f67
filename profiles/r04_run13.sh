#!/bin/bash
# Round 4, thirteenth GPU call: column phases per workgroup of the strip-record tile writer (LM_FE_ROWS_CS = all / 2 / 1): k_fe_bits split under the profiler.
OUT=${1:-gpurun_out/r04u}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
(timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -k "bit_planes" 2>&1 | tail -4) > $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for cs in 0 2 1; do
LM_FE_ROWS_CS=$cs LM_FE_BITS_SPLIT=1 timeout 300 rocprofv3 --kernel-trace -d $ROOT/$OUT/prof$cs -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate --no-pmc > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
DB=$(find $ROOT/$OUT/prof$cs -name "*_results.db" | head -1)
python - <<PY
import sqlite3
con = sqlite3.connect("$DB")
rows = con.execute("select start, end, name from kernels order by start").fetchall()
fe = [(e - s) / 1e3 for s, e, n in rows if "k_fe_bits" in n]
print("LM_FE_ROWS_CS=$cs k_fe_bits launches (us), in order:", [round(x, 1) for x in fe[-12:]])
PY
done
find $ROOT/$OUT -name "*_results.db" -delete
