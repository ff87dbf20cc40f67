#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_run27
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -k "icp or pose or refine or pipeline" 2>&1 | tail -8) > $OUT/pytest.log
cat $OUT/pytest.log
for w in 8 64; do
  echo "LM_KNN_HARD_LANES=$w"
  LM_KNN_HARD_LANES=$w timeout 300 python profiles/icp_only.py 16 2>&1 | grep device_ms | cut -c1-160
  LM_KNN_HARD_LANES=$w timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done > $OUT/legs.txt 2>&1
cat $OUT/legs.txt
TEAM_MEMBERS=0 timeout 300 python profiles/r06_icp_team.py 16 2>&1 | grep "k_icp_knn" | head -2
