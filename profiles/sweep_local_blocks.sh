#!/bin/bash
# Grid size of the persistent refinement kernel (workgroups of 4 waves): ms per frame of the live-stream bench and the
# kernels' own durations.  Usage (GPU box): bash profiles/sweep_local_blocks.sh > gpurun_out/sweep.txt
for lb in ${LM_SWEEP:-768 1024 1280 1536 1792 2048}; do
  LM_LOCAL_BLOCKS=$lb python bench.py --no-extras --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); o=j['roofline']['other']; h=j['host_wall_ms']
print('blocks', $lb, 'ms/frame %.4f' % j['ms_per_step'], 'alone: coarse %.1f us local %.1f us' % (o['k_coarse_ms']*1e3, o['k_local_ms']*1e3), 'pipelined: coarse %.1f local %.1f fe %.1f h2d %.1f' % (j['stages_ms']['coarse_ms']*1e3, j['stages_ms']['local_ms']*1e3, j['stages_ms']['frontend_ms']*1e3, j['stages_ms']['h2d_ms']*1e3), 'host: submit %.3f collect %.3f (lib: submit %.3f wait %.3f collect %.3f merge %.3f)' % (h['submit'], h['collect'], h['host_submit_ms'], h['host_wait_ms'], h['host_collect_ms'], h['host_merge_ms']))"
done
