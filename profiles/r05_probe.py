"""Round 5 probe (GPU): the plan of the bit-plane refinement on the bench workload — run items / singles per frame, and the refinement's time per 8-frame
launch with and without runs (Detector.setPaths) in one process.  python profiles/r05_probe.py"""
import os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "6dpose_amd"))
import bench, synth
import linemod_oracle as lo
import linemodLevelup_pybind as lm
frames = bench.noisy_frames(9)
od = lo.OracleDetector(bench.NFEAT[0], bench.T_LEVELS)
pyr0 = od.quantize_pyramid(*frames[0])
bank = synth.make_planted_bank(1234, 2000, [(p[0], p[1]) for p in pyr0], bench.T_LEVELS, bench.NFEAT)
out = {}
for refine in ("bits", "bits_single"):
    det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
    det.addClassPacked("obj", *bank)
    det.setPaths(refine, "bits")
    det.setBatch(8); det.setBatchQueue(0)
    res = []
    for rep in range(6):
        for f in frames[1:9]: det.submitFrame(list(f), bench.THRESHOLD, ["obj"])
        for _ in range(8):
            det.collect(sort_unique=True)
            tm = det.lastTimings()
        res.append(tm)
    out[refine] = [{k: tm[k] for k in ("local_ms", "coarse_ms", "frontend_ms", "batch_frames", "coarse_candidates", "run_items", "run_singles")} for tm in res[2:]]
print(json.dumps(out, indent=1))
