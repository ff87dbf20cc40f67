# pipeline leg under the ICP knobs: persistent launch on/off, slices per hypothesis
for cfg in "LM_ICP_PERSIST=0" "LM_ICP_PERSIST=1" "LM_ICP_PERSIST=0 LM_ICP_SPLITS=24" "LM_ICP_PERSIST=0 LM_ICP_SPLITS=64" "LM_ICP_PERSIST=1 LM_ICP_SPLITS=24" "LM_ICP_PERSIST=1 LM_ICP_SPLITS=16"; do
  echo "$cfg: $(env $cfg python profiles/pipeline_only.py 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("icp_ms %.3f total_ms %.3f iters %d fitness %.6f" % (d["icp_ms"], d["total_ms"], d["icp_iterations"], d["mean_fitness"]))')"
done
