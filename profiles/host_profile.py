"""Where the host thread's time goes in the live-stream loop (round 3): python-level wall time of submitFrame / collect against the
library's own accounting (lm_detector_host_profile).  GPU box: python profiles/host_profile.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, linemodLevelup_pybind as lm, synth
W, H = bench.W, bench.H
print('host cpus:', None if os.environ.get('LM_NO_BIND') else lm.bind_near_device(0), flush=True)   # as bench.py does (LM_NO_BIND=1: leave the placement to the scheduler)
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(16)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
det.addClassPacked("obj", *synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT))
cls = ["obj"]
for steps in (20, 200):
    for depth in (8, 12):
        def run(n, rec):
            infl = 0; ts = tc = 0.0
            for k in range(n):
                t0 = time.perf_counter(); det.submitFrame(frames[k % 16], 75.0, cls); ts += time.perf_counter() - t0; infl += 1
                if infl == depth:
                    t0 = time.perf_counter(); det.collect(); tc += time.perf_counter() - t0; infl -= 1
            while infl:
                t0 = time.perf_counter(); det.collect(); tc += time.perf_counter() - t0; infl -= 1
            return ts, tc
        run(32, False); det.hostProfile()
        t0 = time.perf_counter(); ts, tc = run(steps, True); dt = time.perf_counter() - t0
        hp = det.hostProfile()
        n = hp["frames"]
        print("steps %d depth %d: %.4f ms/frame; python submit %.1f us collect %.1f us | library per frame (us): %s" % (
            steps, depth, dt / steps * 1e3, ts / steps * 1e6, tc / steps * 1e6, {k: round(v / n * 1e6, 1) for k, v in hp.items() if k != "frames"}), flush=True)
