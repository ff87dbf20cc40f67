"""The `extras.pipeline` leg of bench.py alone (match 2k templates -> NMS -> top-16 -> ICP), for kernel traces:
   rocprofv3 --kernel-trace -d out -- python profiles/pipeline_only.py [steps]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import bench, linemodLevelup_pybind as lm, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(2)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(bench.THRESHOLD, ["_probe"])
quant = [(det.readStage(l, 0).reshape(bench.H >> l, bench.W >> l), det.readStage(l, 1).reshape(bench.H >> l, bench.W >> l)) for l in range(2)]
bank = synth.make_planted_bank(1234, bench.N_TEMPLATES, quant, bench.T_LEVELS, bench.NFEAT)
det.addClassPacked("obj00", *bank)
if os.environ.get("PIPE_DIAG"):
    _close = lm.Pipeline.close
    def close_with_dump(self):
        for hyp in range(16):
            st = self.read_icp_debug(hyp, 3)
            d = self.read_icp_debug(hyp, 4).reshape(2, 64, 32)
            par = int(os.environ.get('PIPE_DIAG_PARITY', '0'))
            tot, sea, wall = d[par, :, 29], d[par, :, 30], d[par, :, 31]
            act = tot > 0
            if not act.any(): continue
            worst = int(np.argmax(tot))
            print("   worst slice %d: cycles %.0f search %.0f max over lanes: (candidates, columns*1000+queued) %s points %d..%d | median slice classes %s" % (worst, tot[worst], sea[worst], divmod(int(wall[worst]), 1000000), 0, 0, sorted(int(w) for w in wall[act])[len(wall[act]) // 2]), file=sys.stderr)
            print("hyp %2d iters %2d n_model %5d n_scene %5d grid %dx%d | slices %d: cycles min %.0f median %.0f max %.0f, search median %.0f max %.0f, wall us max %.1f | not in LDS %d, largest slab %d | corr/slice max %.0f" % (
                hyp, st[24], st[19], st[20], st[21], st[22], act.sum(), tot[act].min(), np.median(tot[act]), tot[act].max(), np.median(sea[act]), sea[act].max(), wall[act].max() / 100, st[31], st[32], d[par, act, 28].max()), file=sys.stderr)
        _close(self)
    lm.Pipeline.close = close_with_dump
print(json.dumps(bench.pipeline_bench(det, frames, bank, ["obj00"], steps=steps)))
