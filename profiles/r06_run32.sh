#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
(timeout 900 python -m pytest tests -m gpu -q -x -k "nms or pipeline or topk" 2>&1 | tail -4)
timeout 300 python profiles/pipeline_only.py 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
