#!/bin/bash
# Round 5, eighth GPU call: T = 5 pixel-tile strip records (parity of every writer, then the real-fixture leg's front end with 1 / 5 column phases per workgroup)
OUT=${1:-gpurun_out/r05h}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_planes_equal or fixture_banks" > $OUT/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.log
for cs in 1 5; do
LM_FE_ROWS_CS=$cs python - <<PY
import sys, os, json
sys.path[:0] = ["$ROOT", "$ROOT/6dpose_amd", "$ROOT/tests"]
import bench
bench.PIPELINE_DEPTH = 16
r = bench.real_fixture_leg(0)
print("cs=$cs", {k: r[k] for k in ("ms_per_frame", "equals_oracle", "frames_per_launch_mean", "kernels_ms_per_frame")})
PY
done
for v in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver flags with gate: ms_per_step', d['ms_per_step'], [s['frames_in_its_launch'] for s in d['parity']['stream']['timed_region_rule']['steps']], d['parity']['stream']['timed_region_rule']['frames_in_the_launch_of_each_step'])"; done
for v in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-parity-gate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver flags no gate: ms_per_step', d['ms_per_step'])"; done
