#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python profiles/r06_sort_clk.py 16 2>&1 | grep -v amdgpu.ids
