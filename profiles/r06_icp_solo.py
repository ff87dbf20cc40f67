"""Round 6: k_icp_solo (one workgroup per hypothesis, all evaluations in one launch) against the sliced launches of rounds 1-5.
LM_ICP_SOLO = -1 (sliced only), 0 (all evaluations in k_icp_solo), K > 0 (K sliced evaluations first).  Prints the icp leg of
bench.py per setting, the agreement of the poses with the sliced path, and the phase split of wave 0 (shader cycles)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import bench, synth
import linemodLevelup_pybind as lm

def poses(solo, hypotheses=16):
    os.environ["LM_ICP_SOLO"] = str(solo)
    K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
    rng = np.random.default_rng(7)
    scene_model = synth.synth_model_depth(100)
    scene = np.where(scene_model > 0, scene_model + 4, 0).astype(np.uint16)
    scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
    mds, xy = [], []
    for h in range(hypotheses):
        md = synth.synth_model_depth(100 + (h % 4))
        ys, xs = np.nonzero(md)
        mds.append(md)
        xy.append((int(xs.min()) + int(rng.integers(-2, 3)), int(ys.min()) + int(rng.integers(-2, 3))))
    Ks = np.tile(K.reshape(1, 9), (hypotheses, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (hypotheses, 1))
    ts = np.tile(np.array([[0, 0, 1000]], np.float32), (hypotheses, 1))
    ctx = lm.IcpContext(device=0, scene_from_scene=True)
    ctx.set_scene(scene, K); ctx.set_models(mds)
    res, ms = ctx.run(Ks, Rs, ts, xy)
    dbg = [ctx.read_debug(h, 3) for h in range(hypotheses)]
    per_eval = [ctx.read_debug(h, 4).reshape(2, 64, 32)[0] for h in range(hypotheses)]
    ctx.close()
    return res, dbg, per_eval

ref, _, _ = poses(-1)
for solo in [int(a) for a in sys.argv[1:]] or [-1, 0, 1, 2, 3]:
    os.environ["LM_ICP_SOLO"] = str(solo)
    out = bench.icp_bench(0, hypotheses=16, reps=5)
    res, dbg, per_eval = poses(solo)
    dR = max(np.abs(a["R"] - b["R"]).max() for a, b in zip(res, ref)); dt = max(np.abs(a["t"] - b["t"]).max() for a, b in zip(res, ref))
    its = [r["iterations"] for r in res]
    print(json.dumps({"LM_ICP_SOLO": solo, "device_ms": out["device_ms"], "wall_ms": out["wall_ms"], "iterations_total": out["iterations_total"],
                      "iters_per_sec_device": out["icp_iters_per_sec_device"], "max_dR_vs_sliced": dR, "max_dt_mm_vs_sliced": dt,
                      "iterations_equal": its == [r["iterations"] for r in ref], "iterations": its,
                      "n_source": [r["n_source"] for r in res], "n_target": [r["n_target"] for r in res]}))
    if solo >= 0:
        for h, d in enumerate(dbg):
            clk = d[25:33]
            if clk[5] > 0:
                print("  hyp %2d evals %2d: wave-0 cycles per evaluation: finish %.0f, transform+queue %.0f, search %.0f, sums %.0f | kernel %.0f cycles, queued/eval %.1f, queue capacity %d" % (
                    h, clk[5], clk[0] / clk[5], clk[1] / clk[5], clk[3] / clk[5], clk[4] / clk[5], clk[2], clk[6] / clk[5], clk[7]))
                if solo == 0 and h in (1, 3):
                    for it in range(int(clk[5])):
                        r = per_eval[h][it]
                        print("      eval %2d: searches %4d, motion bound so far %.3f mm, cycles finish %6d transform+queue %6d search %7d sums %6d" % (it, r[0], r[1] * 1e3, r[5], r[2], r[3], r[4]))
