# full GPU check of the tree: parity suite, smoke, default bench line, kernel trace of the bench (stats)
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" ; tail -2 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
( time python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err ) 2> gpurun_out/final/bench.time; tail -3 gpurun_out/final/bench.time | head -1
python -c "import json; d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'icp', d['extras']['icp']['device_ms'], 'pipeline', d['extras']['pipeline']['total_ms'], d['extras']['pipeline']['icp_ms'], 'cpu', d['cpu_baseline']['value'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -o v5 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final/prof.log 2>&1; echo "rocprof rc=$?"
ls $GRAFT_REPO_ROOT/gpurun_out/final/prof | head
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_flags.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/final/bench_driver_flags.json').read().strip().splitlines()[-1]); print('driver flags: ms_per_step', d['ms_per_step'], 'pipeline', d['extras']['pipeline']['total_ms'], 'icp', d['extras']['icp']['device_ms'])"
