"""The driver times `bench.py --steps 20 --warmup 5`: a timed region that starts and ends with an empty pipeline.  Median over repeated
20-step regions (same process, same GPU) for the knobs that shape pipeline fill and drain: frames per launch, launched batches kept
queued, frames in flight.  GPU box: python profiles/short_run_sweep.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, linemodLevelup_pybind as lm, synth
import torch
W, H = bench.W, bench.H
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(16)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
det.addClassPacked("obj", *synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT))
cls = ["obj"]

def run(n, depth):
    infl = 0
    for k in range(n):
        det.submitFrame(frames[k % 16], 75.0, cls); infl += 1
        if infl == depth:
            det.collect(); infl -= 1
    while infl:
        det.collect(); infl -= 1

run(32, 8)
steps = int(os.environ.get("STEPS", "20"))
for batch in (1, 2, 4, 8):
    det.setBatch(batch)
    for queue in (1, 2, 3):
        det.setBatchQueue(queue)
        row = []
        for depth in (4, 8, 12, 16):
            ts = []
            for rep in range(15):
                run(5, depth)
                torch.cuda.synchronize()
                t0 = time.perf_counter(); run(steps, depth); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps * 1e3)
            row.append("depth %2d: %.4f (min %.4f)" % (depth, float(np.median(ts)), min(ts)))
        print("batch %d queue %d | %s" % (batch, queue, " | ".join(row)), flush=True)
