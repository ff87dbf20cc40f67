# timing experiments on k_coarse's planner (LM_COARSE_DBG: 1 = no grouping, 2 = no global atomic; results are wrong with 2)
for cfg in "0 8" "0 4" "0 2" "0 1" "1 8" "2 8" "3 8"; do
  set -- $cfg
  echo "LM_COARSE_DBG=$1 LM_COARSE_GROUP=$2"; LM_COARSE_DBG=$1 LM_COARSE_GROUP=$2 LM_SWEEP=2048 bash profiles/sweep_local_blocks.sh 2>&1
done
