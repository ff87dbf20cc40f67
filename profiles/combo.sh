# coarse templates per workgroup x refinement grid x frames in flight
for depth in 3 4; do for grp in 1 2 4; do for lb in 1536 2048 4096 8192; do
  echo -n "depth $depth group $grp "; LM_BENCH_DEPTH=$depth LM_COARSE_GROUP=$grp LM_SWEEP=$lb bash profiles/sweep_local_blocks.sh 2>&1 | cut -c1-150
done; done; done
