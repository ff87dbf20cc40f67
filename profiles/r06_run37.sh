#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
(timeout 1500 python -m pytest tests -m gpu -q -x -k "not icp and not pose and not pipeline and not rccl and not sharded and not bench" 2>&1 | tail -4)
timeout 600 python bench.py --steps 50 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.4f kernel us/frame %.1f' % (d['ms_per_step'], d['kernel_us_per_frame']), 'coarse ms', r['other']['k_coarse_ms'], 'local', r['other']['k_local_ms'])
print({k:v.get('ceilings') for k,v in r['stages'].items() if isinstance(v,dict) and 'ceilings' in v})"
timeout 600 python bench.py --scaling strong --steps 50 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('16k: ms/step %.4f' % d['ms_per_step'], 'coarse ms', r['other']['k_coarse_ms'], 'local', r['other']['k_local_ms'])"
