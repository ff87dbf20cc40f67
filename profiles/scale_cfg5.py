"""One GPU's share of BASELINE configs[4]: 1280x960 RGB-D, 90k templates / 8 GPUs = 11250 template pyramids per GPU: stage split
and rates (the bit-exact check of this size against the CPU oracle is tests/test_gpu_parity.py::test_config4_shard_size)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, synth
W, H, T, NF, N = 1280, 960, [4, 8], (150, 75), int(sys.argv[1]) if len(sys.argv) > 1 else 11250
rgb, dep = synth.make_frame(0, W, H)
det = lm.Detector(NF[0], T, device=0)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame([rgb, dep]); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
bank = synth.make_planted_bank(99, N, quant, T, NF)
det.addClassPacked("obj", *bank)
det.storeFrame(0, (rgb, dep))
for _ in range(3):
    det.selectFrame(0); out = det.matchResident(75.0, ["obj"])
t0 = time.perf_counter()
for _ in range(10):
    det.selectFrame(0); out = det.matchResident(75.0, ["obj"])
dt = (time.perf_counter() - t0) / 10
tm = det.lastTimings()
res = {"frame": [W, H], "templates": N, "ms_per_frame_sync": dt * 1e3, "value_templates_Mpx_per_s": N * W * H / 1e6 / dt,
       "stages_ms": {k: tm[k] for k in ("frontend_ms", "coarse_ms", "local_ms")}, "coarse_candidates": tm["coarse_candidates"],
       "matches_pre_unique": tm["matches_pre_unique"], "matches": len(out),
       "k_coarse_GBps": tm["coarse_bytes"] / (tm["coarse_ms"] * 1e-3) / 1e9, "k_local_GBps": tm["local_bytes"] / (tm["local_ms"] * 1e-3) / 1e9}
print(json.dumps(res))
