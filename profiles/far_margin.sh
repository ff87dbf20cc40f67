cd $GRAFT_REPO_ROOT/6dpose_amd/csrc
for fm in 1.5 1.3 1.2 1.1; do
  sed -i "s/^constexpr double kFarMargin = [0-9.]*;/constexpr double kFarMargin = $fm;/" icp.hip
  make > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "far margin $fm: $(python profiles/pipeline_only.py 10 2>&1 | grep match_ms | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"icp_ms\"],3), d[\"icp_iterations\"], round(d[\"mean_fitness\"],6))") | $(python profiles/icp_only.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"device_ms\"],3), d[\"iterations_total\"])")"
  cd $GRAFT_REPO_ROOT/6dpose_amd/csrc
done
sed -i "s/^constexpr double kFarMargin = [0-9.]*;/constexpr double kFarMargin = 1.5;/" icp.hip
