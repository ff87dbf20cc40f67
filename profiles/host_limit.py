"""Where the frame time of the live-stream loop goes: the same 2000-template workload with the host-side pieces removed one
by one (staging copy, canonical sort), at 3 and 4 frames in flight.  GPU box: python profiles/host_limit.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, linemodLevelup_pybind as lm, synth
W, H = bench.W, bench.H
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(8)
for k in range(4):
    det.storeFrame(k, frames[k])
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.selectFrame(0); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
det.addClassPacked("obj", *synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT))
cls = ["obj"]

def loop(mode, depth, steps=200, warm=20):
    def sub(k):
        if mode == "pageable": det.submitFrame(frames[k % 8], 75.0, cls)
        elif mode in ("pinned", "pinned_nosort"):
            r, d = det.ingestBuffers(W, H); det.submitFrame((r, d), 75.0, cls)     # the caller left the frame in the pinned ring
        else: det.selectFrame(k % 4); det.submit(75.0, cls)
    def col():
        if mode.endswith("nosort"): det.collect(sort_unique=False, distinct=True)
        else: det.collect()
    def run(n):
        infl = 0
        for k in range(n):
            sub(k); infl += 1
            if infl == depth: col(); infl -= 1
        while infl: col(); infl -= 1
    run(warm)
    t0 = time.perf_counter(); run(steps); return (time.perf_counter() - t0) / steps * 1e3

for batch in [int(b) for b in os.environ.get("BATCHES", "4").split(",")]:
    det.setBatch(batch)
    for depth in [int(x) for x in os.environ.get("DEPTHS", "8,12,16").split(",")]:
        print("batch", batch, "depth", depth, " ".join("%s %.4f" % (m, loop(m, depth)) for m in os.environ.get("MODES", "pageable,pinned,pinned_nosort").split(",")), flush=True)
