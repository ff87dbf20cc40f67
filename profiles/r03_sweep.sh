#!/bin/bash
# Round 3: batch / queue sweep of the headline loop + rocprofv3 kernel stats of the roofline leg.  Usage (GPU box): profiles/r03_sweep.sh <out_dir>
OUT=${1:-gpurun_out/r03c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
for q in 1 2 3; do for st in 20 200; do
  timeout 200 python bench.py --no-extras --no-cpu-baseline --no-parity-gate --batch-queue $q --steps $st --warmup 5 > $OUT/bench_q${q}_s$st.json 2> $OUT/bench_q${q}_s$st.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_q${q}_s$st.json"))
print("queue $q steps $st ms/step %.4f  mean batch %.2f  host %s" % (d["ms_per_step"], d["config"]["frames_per_launch_mean_timed"], {k: round(v,3) for k,v in d["host_wall_ms"].items() if k in ("submit","collect")}))
PY
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o roof -- python $ROOT/bench.py --roofline-only --no-parity-gate > $ROOT/$OUT/roofline_only.json 2> $ROOT/$OUT/roofline_only.err
cat $ROOT/$OUT/roofline_only.json
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/kernel_stats_roofline_leg.txt | head -20
