"""GPU check of the bit-plane refinement prototype (make -C 6dpose_amd/csrc BITS=1): the bench workload's frames 0 and 1 through
Detector.match with LM_BITPLANES=1 against the CPU oracle.  AMD_LINEMOD_LIB=6dpose_amd/libamdlinemod_bits.so LM_BITPLANES=1 python profiles/bitplane_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench, linemodLevelup_pybind as lm, synth
import linemod_oracle as lo
W, H = bench.W, bench.H
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(4)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
bank = synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT)
det.addClassPacked("obj", *bank)
od = lo.OracleDetector(bench.NFEAT[0], bench.T_LEVELS)
pb = lo.PackedBank(2000, 2, *bank)
ok = True
for k in range(2):
    rgb, dep = frames[k]
    want, _, st, _, _, _ = bench.oracle_matches(od, lo, pb, rgb, dep, 75.0)
    got = det.matchArray([rgb, dep], 75.0, ["obj"])
    tm = det.lastTimings()
    same = bench.same_records(got, want)
    ok = ok and same
    print("frame %d: gpu %d records, oracle %d, equal %s; candidates %d / %d, evals %d, local_ms %.3f" % (
        k, len(got), len(want), same, int(tm["coarse_candidates"]), int(st["coarse_candidates"]), int(tm["local_evals"]), tm["local_ms"]), flush=True)
    if not same:
        g = set(zip(got["x"].tolist(), got["y"].tolist(), got["similarity"].tolist(), got["template_id"].tolist()))
        w = set(zip(want["x"].tolist(), want["y"].tolist(), want["sim"].tolist(), want["tid"].tolist()))
        print("  only gpu %d, only oracle %d; samples gpu-only %s oracle-only %s" % (len(g - w), len(w - g), sorted(g - w)[:4], sorted(w - g)[:4]), flush=True)
print("BITPLANES PARITY", "OK" if ok else "FAILED")
sys.exit(0 if ok else 3)
