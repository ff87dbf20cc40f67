"""Round 6: which hypotheses of the pipeline leg k_icp_team leaves to the sliced launches, and why (IcpState::team_note)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import bench, linemodLevelup_pybind as lm, synth
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(2)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(bench.THRESHOLD, ["_probe"])
quant = [(det.readStage(l, 0).reshape(bench.H >> l, bench.W >> l), det.readStage(l, 1).reshape(bench.H >> l, bench.W >> l)) for l in range(2)]
bank = synth.make_planted_bank(1234, bench.N_TEMPLATES, quant, bench.T_LEVELS, bench.NFEAT)
det.addClassPacked("obj00", *bank)
_close = lm.Pipeline.close
def close_with_dump(self):
    for hyp in range(16):
        st = self.read_icp_debug(hyp, 3)
        print("hyp %2d iterations %2d grid %dx%d team_note %s n_src %d n_tgt %d | voxel (model cloud of %d points): extent %d keys %d sort %d means %d cycles | knn: staging %d trips %d whole-wave %d, %d hard" % (
            hyp, st[24], st[21], st[22], [int(x) for x in st[33:37]], st[37], st[38], st[47], st[43], st[44], st[45], st[46], st[39], st[40], st[41], st[42]))
    _close(self)
lm.Pipeline.close = close_with_dump
print(json.dumps(bench.pipeline_bench(det, frames, bank, ["obj00"], steps=2))[:300])
