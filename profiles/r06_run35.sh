#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
(timeout 1200 python -m pytest tests -m gpu -q -x -k "preparation_paths or more_hypotheses or icp_context or batch_equals" --durations=4 2>&1 | tail -12)
