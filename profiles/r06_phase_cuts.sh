#!/bin/bash
# Round 6: a cramped batch leaves k_icp_team's first launch after evaluation LM_ICP_CUT_INDEX and the chip is dealt out again (LM_ICP_RELAUNCH launches more):
# ICP / pipeline tests, then both legs with and without, then which hypotheses were suspended where (profiles/r06_pipe_team.py)
cd "$GRAFT_REPO_ROOT" || exit 1
(timeout 900 python -m pytest tests -m gpu -q -x -k "icp or pose or refine or pipeline" 2>&1 | tail -3)
for c in "0 3" "1 3"; do
  set -- $c
  echo "LM_ICP_RELAUNCH=$1 LM_ICP_CUT_INDEX=$2"
  for r in 1 2; do LM_ICP_RELAUNCH=$1 LM_ICP_CUT_INDEX=$2 timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-100; done
  LM_ICP_RELAUNCH=$1 LM_ICP_CUT_INDEX=$2 timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
done
timeout 300 python profiles/r06_pipe_team.py 2>&1 | grep "^hyp" | sed "s/ | member 0: / | /; s/per evaluation:.*| note/| note/"
