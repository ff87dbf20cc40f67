for cfg in "LM_ICP_SPLITS=48" "LM_ICP_SPLITS=32" "LM_ICP_SPLITS=64"; do
  echo "$cfg: $(env $cfg PIPE_DIAG=1 python profiles/pipeline_only.py 10 2>&1 | grep -E "hyp  0|match_ms" | sed -e 's/"wall_ms.*//' -e 's/n_model.*slices/slices/')"
done
python profiles/icp_only.py 2>&1 | tail -2
