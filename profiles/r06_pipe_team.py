"""Round 6: the pipeline leg's 16 hypotheses in k_icp_team — iterations, cloud sizes, the kernel's cycles and phase split of each hypothesis' member 0
(IcpState::clk as profiles/r06_icp_team.py reads it), to see which hypothesis sets the length of the launch."""
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
        clk = st[25:33]
        ev = max(clk[5], 1)
        print("hyp %2d iterations %2d n_src %5d n_tgt %5d grid %2dx%2d | member 0: kernel %7d cycles, %2d evaluations, per evaluation: exchange %5d finish %5d move+queue %5d search %5d sums %5d, own searches %.0f | note %s | last team %d workgroups, suspended at evaluation %d" % (
            hyp, st[24], st[37], st[38], st[21], st[22], clk[2], clk[5], clk[7] / ev, (clk[0] - clk[7]) / ev, clk[1] / ev, clk[3] / ev, clk[4] / ev, clk[6] / ev, [int(x) for x in st[33:37]], st[65], st[66]))
    _close(self)
lm.Pipeline.close = close_with_dump
print(json.dumps(bench.pipeline_bench(det, frames, bank, ["obj00"], steps=2))[:200])
