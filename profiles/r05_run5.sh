#!/bin/bash
# Round 5, fifth GPU call: ICP — launch per evaluation against the persistent kernel (LM_ICP_PERSIST=1), kernel trace of the 16-hypothesis leg
OUT=${1:-gpurun_out/r05e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
for p in 0 1; do
  LM_ICP_PERSIST=$p timeout 300 python profiles/icp_only.py 16 > $OUT/icp_p$p.json 2> $OUT/icp_err$p.txt
  python -c "
import json; d=json.loads(open('$OUT/icp_p$p.json').read().strip().splitlines()[-1]); print('persist=$p icp16', {k: d[k] for k in ('device_ms','wall_ms','iterations_total') if k in d})"
  LM_ICP_PERSIST=$p timeout 300 python profiles/pipeline_only.py > $OUT/pipe_p$p.json 2>> $OUT/icp_err$p.txt
  tail -1 $OUT/pipe_p$p.json | cut -c1-400
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o icp -- python $ROOT/profiles/icp_only.py 16 > /dev/null 2> $ROOT/$OUT/err.txt
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
python $ROOT/profiles/rocpd_summary.py $DB $ROOT/$OUT/stats_icp.txt > /dev/null
grep -E "k_icp" $ROOT/$OUT/stats_icp.txt | cut -c1-40,60-150
find $ROOT/$OUT -name "*_results.db" -delete
