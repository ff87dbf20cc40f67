#!/bin/bash
# Kernel timeline of ONE timed region of `bench.py --steps 20 --warmup 5` (the driver's flags): where the GPU idles.  Usage: profiles/r04_trace20.sh [out_dir]
OUT=${1:-gpurun_out/r04trace}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $ROOT/$OUT/prof -o t20 -- python $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate > $ROOT/$OUT/bench.json 2> $ROOT/$OUT/bench.err
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
python - <<PY
import sqlite3, json
con = sqlite3.connect("$DB")
rows = con.execute("select start, end, name from kernels order by start").fetchall()
names = [r[2].split("(")[0].replace("lm::", "").replace("void ", "")[:18] for r in rows]
# the timed region = the last 20 frames: walk back from the end until 20 frames' worth of k_dedupe launches (grid.y frames each) - simply the last 40 kernels
tail = rows[-34:]
t0 = tail[0][0]; prev = tail[0][0]
out = []
for (s, e, n) in tail:
    nm = n.split("(")[0].replace("lm::", "").replace("void ", "")[:16]
    out.append("%-16s start %8.1f dur %7.1f gap %7.1f" % (nm, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
open("$ROOT/$OUT/timeline_last_region.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
try:
    d = json.load(open("$ROOT/$OUT/bench.json")); print("ms/step under the profiler %.4f" % d["ms_per_step"])
except Exception as ex:
    print("bench json:", ex)
try:
    cp = con.execute("select start, end, bytes from memory_copies order by start").fetchall()
    print("memory copies:", len(cp), "last 22 (start rel to first kernel of the window, dur, bytes):")
    for s, e, b in cp[-22:]:
        print("  copy start %8.1f dur %6.1f bytes %d" % ((s - t0) / 1e3, (e - s) / 1e3, b))
except Exception as ex:
    print("copies:", ex)
PY
find $ROOT/$OUT -name "*_results.db" -delete
