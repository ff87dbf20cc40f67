"""Timeline of ONE 20-step region of the live-stream loop (as bench.py --steps 20 times it): when each submit returns, when each collect
returns, how many frames the launch served.  GPU box: python profiles/short_run_timeline.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, linemodLevelup_pybind as lm, synth
import torch
W, H = bench.W, bench.H
print('host cpus:', None if os.environ.get('LM_NO_BIND') else lm.bind_near_device(0), flush=True)   # as bench.py does (LM_NO_BIND=1: leave the placement to the scheduler)
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(16)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
det.addClassPacked("obj", *synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT))
cls = ["obj"]
depth = int(os.environ.get("DEPTH", "16"))
def tm(det):
    t = det.lastTimings()
    return "%d fe%.0f co%.0f lo%.0f tot%.0f" % (t["batch_frames"], t["frontend_ms"] * 1e3, t["coarse_ms"] * 1e3, t["local_ms"] * 1e3, t["total_ms"] * 1e3)
def run(n, log=None):
    infl = 0
    t0 = time.perf_counter()
    for k in range(n):
        det.submitFrame(frames[k % 16], 75.0, cls); infl += 1
        if log is not None: log.append(("S%d" % k, (time.perf_counter() - t0) * 1e3, det.framesLaunched()))
        if infl == depth:
            det.collect(); infl -= 1
            if log is not None: log.append(("C", (time.perf_counter() - t0) * 1e3, tm(det)))
    while infl:
        det.collect(); infl -= 1
        if log is not None: log.append(("C", (time.perf_counter() - t0) * 1e3, tm(det)))
run(32)
for rep in range(3):
    run(5); torch.cuda.synchronize()
    base = det.framesLaunched()
    log = []
    run(20, log); torch.cuda.synchronize()
    print("rep", rep, "total %.3f ms" % log[-1][1])
    print("  " + " ".join("%s@%.2f(%s)" % (a, t, (c - base) if a[0] == "S" else c) for a, t, c in log))
