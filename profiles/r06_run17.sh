#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_flags.json 2> gpurun_out/r06_bench_driver_flags.err ) 2> gpurun_out/r06_bench_time.txt
tail -3 gpurun_out/r06_bench_driver_flags.err; cat gpurun_out/r06_bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_driver_flags.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "repeats", "ms_per_step_steady_200", "kernel_us_per_frame", "speedup_vs_cpu_1thread", "speedup_vs_cpu_1thread_incl_quantisation", "parity_checked")})
ex = d["extras"]
print("icp", {k: ex["icp"][k] for k in ("device_ms", "iterations_total", "icp_iters_per_sec_device")})
print("pipeline", {k: ex["pipeline"][k] for k in ("match_ms", "nms_ms", "icp_ms", "total_ms")})
print("roofline_icp", {k: ex["roofline_icp"][k] for k in ("achieved", "peak", "frac", "vs_f64_vector_peak")})
print("proxy", ex["strong_scaling_proxy"]["ratio"])
PY
