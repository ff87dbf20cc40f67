"""ICP-only driver for rocprofv3 (profiles/): runs bench.icp_bench on cuda:0 and prints its summary."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
hyp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print(json.dumps(bench.icp_bench(0, hypotheses=hyp, reps=5)))
