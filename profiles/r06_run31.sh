#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python bench.py --steps 50 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.4f kernel us/frame %.1f' % (d['ms_per_step'], d['kernel_us_per_frame']))
print(json.dumps({k:v for k,v in d['config'].items() if 'ms' in k or 'stage' in k})[:600])
print(json.dumps(r)[:1500])"
