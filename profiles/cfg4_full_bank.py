"""BASELINE configs[4] as a whole on ONE GPU: 30 objects x 3000 template pyramids = 90k templates, 1280x960 RGB-D stream.
The bank is resident (90k x 450 features: 0.65 GB of the 288 GB); a frame is one Detector.match over all 30 classes, frames come
from host memory through the live-stream ingest, 4 in flight.  Also the 8-GPU split's per-rank share (objects 0..3 of 30: the
largest whole-object share when 30 objects are dealt to 8 ranks) for the strong-scaling estimate.  GPU box:
python profiles/cfg4_full_bank.py [objects=30] [templates_per_object=3000]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, synth
W, H, T, NF = 1280, 960, [4, 8], (150, 75)
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 30
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
frames = [synth.make_frame(s, W, H, 80) for s in range(3)]
det = lm.Detector(NF[0], T, device=0)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
t0 = time.perf_counter()
classes = []
for o in range(n_obj):
    det.addClassPacked("obj%02d" % o, *synth.make_planted_bank(500 + o, per, quant, T, NF))
    classes.append("obj%02d" % o)
t_bank = time.perf_counter() - t0

def stream(cls, steps, depth=4, warm=4):
    acc, n = {}, 0
    def go(k0, cnt, rec):
        nonlocal n
        infl = 0
        for k in range(k0, k0 + cnt):
            det.submitFrame(frames[k % len(frames)], 75.0, cls); infl += 1
            if infl == depth:
                out = det.collect(); infl -= 1
                if rec:
                    n += 1
                    for q, v in det.lastTimings().items(): acc[q] = acc.get(q, 0.0) + v
        while infl: out = det.collect(); infl -= 1
        return out
    det.matchArray(list(frames[0]), 75.0, cls)             # grows the candidate buffers to this bank's size
    go(0, warm, False)
    t = time.perf_counter(); out = go(warm, steps, True); dt = (time.perf_counter() - t) / steps
    return dt, {q: v / max(1, n) for q, v in acc.items()}, len(out)

res = {"frame": [W, H], "objects": n_obj, "templates_per_object": per, "bank_build_s": t_bank}
# parity on this configuration's own frame: one object's 3000 templates at 1280x960 against the CPU oracle (numpy quantisation +
# oracle/match_oracle.c), record by record and the coarse candidate count
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import bench, linemod_oracle as lo
od = lo.OracleDetector(NF[0], T)
pb = lo.PackedBank(per, 2, *synth.make_planted_bank(500, per, quant, T, NF))
want, _, st, _, _, _ = bench.oracle_matches(od, lo, pb, frames[0][0], frames[0][1], 75.0)
got = det.matchArray(list(frames[0]), 75.0, [classes[0]])
res["parity"] = {"class": classes[0], "matches": int(len(want)), "coarse_candidates": int(st["coarse_candidates"]),
                 "equal": bool(bench.same_records(got, want) and int(det.lastTimings()["coarse_candidates"]) == int(st["coarse_candidates"]))}
assert res["parity"]["equal"], res["parity"]
for name, cls in (("whole_bank", classes), ("rank_share_8gpu", classes[:max(1, (n_obj + 7) // 8)])):
    dt, tm, nm = stream(cls, 12 if len(cls) > 8 else 30)
    nt = per * len(cls)
    res[name] = {"templates": nt, "ms_per_frame": dt * 1e3, "value_templates_Mpx_per_s": nt * W * H / 1e6 / dt, "matches_last_frame": nm,
                 "coarse_candidates": tm["coarse_candidates"], "matches_pre_unique": tm["matches_pre_unique"],
                 "stages_ms": {k: tm[k] for k in ("h2d_ms", "frontend_ms", "coarse_ms", "local_ms")},
                 "k_coarse_GBps": tm["coarse_bytes"] / (tm["coarse_ms"] * 1e-3) / 1e9 if tm["coarse_ms"] else None,
                 "k_local_GBps": tm["local_bytes"] / (tm["local_ms"] * 1e-3) / 1e9 if tm["local_ms"] else None}
print(json.dumps(res))
