# kernel traces: durations of the ICP preparation kernels of the last frame (pipeline leg) and of the ICP-only leg
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/tp; cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tp/t $GRAFT_REPO_ROOT/gpurun_out/tp/t2
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tp/t -o tp -- python $GRAFT_REPO_ROOT/profiles/pipeline_only.py 3 > $GRAFT_REPO_ROOT/gpurun_out/tp/run.log 2>&1
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tp/t2 -o tp -- python $GRAFT_REPO_ROOT/profiles/icp_only.py > $GRAFT_REPO_ROOT/gpurun_out/tp/run2.log 2>&1
cd $GRAFT_REPO_ROOT; python profiles/trace_order.py gpurun_out/tp/t | grep -E "k_icp_(voxel|grid|knn|normals|points)"
python profiles/trace_order.py gpurun_out/tp/t | grep k_icp_eval | awk '{s+=$6; n++} END {print "pipeline k_icp_eval launches", n, "total us", s}'
python profiles/trace_order.py gpurun_out/tp/t2 k_icp_bbox | grep -E "k_icp_(voxel|grid|knn|normals|points)"
python profiles/trace_order.py gpurun_out/tp/t2 k_icp_bbox | grep k_icp_eval | awk '{s+=$6; n++} END {print "icp_only k_icp_eval launches", n, "total us", s}'
