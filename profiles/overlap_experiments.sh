# experiments on how the stages of neighbouring frames share the GPU (live-stream bench, 4 frames in flight)
run() { echo -n "$* : "; env "$@" LM_SWEEP=4096 bash profiles/sweep_local_blocks.sh 2>&1 | cut -c1-175; }
run LM_X=base
run LM_FE_FUSED_PIPE=1
run LM_CU_SPLIT=32
run LM_CU_SPLIT=64
run LM_CU_SPLIT=96
run LM_CU_SPLIT=64 LM_FE_FUSED_PIPE=1
run LM_STREAM_PRIO=2210
run LM_STREAM_PRIO=2200
run LM_STREAM_PRIO=0000
