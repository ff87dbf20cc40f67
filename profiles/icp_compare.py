"""ICP legs of bench.py only (configs[2]): 16 hypotheses resident (lm_icp_run) and the match -> NMS -> top-16 -> ICP pipeline.
LM_ICP_PERSIST=0 gives the launch-per-round variant.  GPU box: python profiles/icp_compare.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, linemodLevelup_pybind as lm, synth
W, H = bench.W, bench.H
out = {"persist": os.environ.get("LM_ICP_PERSIST", "1"), "icp": bench.icp_bench(0)}
if "--pipeline" in sys.argv:
    det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
    frames = bench.noisy_frames(2)
    det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
    quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
    bank = synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT)
    det.addClassPacked("obj00", *bank)
    out["pipeline"] = bench.pipeline_bench(det, frames, bank, ["obj00"])
print(json.dumps(out))
