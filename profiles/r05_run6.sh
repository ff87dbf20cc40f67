#!/bin/bash
# Round 5, sixth GPU call: timeline of one ICP run (16 hypotheses): starts and durations of the 32 dependent k_icp_eval launches; phase split
OUT=${1:-gpurun_out/r05f}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $ROOT/$OUT/prof -o icp -- python $ROOT/profiles/icp_only.py 16 > /dev/null 2> $ROOT/$OUT/err.txt
DB=$(find $ROOT/$OUT/prof -name "*_results.db" | head -1)
python $ROOT/profiles/rocpd_timeline.py $DB 0.85 45 > $ROOT/$OUT/timeline.txt
cat $ROOT/$OUT/timeline.txt
find $ROOT/$OUT -name "*_results.db" -delete
cd $ROOT && python profiles/icp_phases.py 2>&1 | tail -18
