"""Per-frame host cost of submitFrame / collect over a 200-step region of the live-stream loop, after the fence bench.py uses: does the start of a
region cost more per frame than its steady state (the driver's 20-step run is all start)?  GPU box: python profiles/submit_series.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, linemodLevelup_pybind as lm, synth
import torch
W, H = bench.W, bench.H
print('host cpus:', None if os.environ.get('LM_NO_BIND') else lm.bind_near_device(0), flush=True)
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(16)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
det.addClassPacked("obj", *synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT))
cls = ["obj"]
def run(n, ts=None, tc=None, depth=16):
    infl = 0
    for k in range(n):
        t0 = time.perf_counter(); det.submitFrame(frames[k % 16], 75.0, cls); t1 = time.perf_counter(); infl += 1
        if ts is not None: ts.append(t1 - t0)
        if infl == depth:
            t0 = time.perf_counter(); det.collect(); t1 = time.perf_counter(); infl -= 1
            if tc is not None: tc.append(t1 - t0)
    while infl:
        t0 = time.perf_counter(); det.collect(); t1 = time.perf_counter(); infl -= 1
        if tc is not None: tc.append(t1 - t0)
run(32)
for label, pre in (("after a fence", None), ("after a fence + 5 ms of spinning", 5e-3), ("after a fence + 50 ms of sleep", -50e-3)):
    for rep in range(2):
        run(5); torch.cuda.synchronize()
        if pre is not None:
            if pre > 0:
                t_end = time.perf_counter() + pre
                while time.perf_counter() < t_end: pass
            else:
                time.sleep(-pre)
        ts, tc = [], []
        t0 = time.perf_counter(); run(200, ts, tc); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        hp = det.hostProfile()
        us = lambda a: "%.1f" % (1e6 * float(np.median(a))) if len(a) else "-"
        print("%s, rep %d: 200 frames %.4f ms/frame; median submit us of frames 0-19 / 20-49 / 50-99 / 100-199: %s / %s / %s / %s; collect: %s / %s / %s / %s" % (
            label, rep, dt / 200 * 1e3, us(ts[:20]), us(ts[20:50]), us(ts[50:100]), us(ts[100:]), us(tc[:20]), us(tc[20:50]), us(tc[50:100]), us(tc[100:])), flush=True)
        print("   submit us, frames 0-23:", " ".join("%.0f" % (1e6 * x) for x in ts[:24]), flush=True)
