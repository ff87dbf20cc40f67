"""Round 6: k_icp_team (RegistrationICP as one launch: a team of workgroups per hypothesis) against the sliced launches of rounds 1-5.
usage: LM_ICP_TEAM=<workgroups per hypothesis> python profiles/r06_icp_team.py [hypotheses]
Prints the icp leg of bench.py for the sliced launches (LM_ICP_SLICED=1) and for the team kernel, the agreement of the poses, and the
phase split of workgroup 0's wave 0 (shader cycles; wave 0 waits at the barriers for the other waves)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import bench, synth
import linemodLevelup_pybind as lm

hyp = int(sys.argv[1]) if len(sys.argv) > 1 else 16

def poses(sliced, hypotheses):
    os.environ["LM_ICP_SLICED"] = "1" if sliced else "0"
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
    dbg4 = [ctx.read_debug(h, 4).reshape(2, 64, 32) for h in range(hypotheses)]
    rows = [d[0, 32:] for d in dbg4]
    global members
    members = [d[1] for d in dbg4]
    ctx.close()
    return res, dbg, rows

ref, _, _ = poses(True, hyp)
for sliced in (True, False):
    os.environ["LM_ICP_SLICED"] = "1" if sliced else "0"
    out = bench.icp_bench(0, hypotheses=hyp, reps=5)
    res, dbg, rows = poses(sliced, hyp)
    dR = max(np.abs(a["R"] - b["R"]).max() for a, b in zip(res, ref)); dt = max(np.abs(a["t"] - b["t"]).max() for a, b in zip(res, ref))
    its = [r["iterations"] for r in res]
    print(json.dumps({"path": "sliced launches" if sliced else "k_icp_team", "LM_ICP_TEAM": os.environ.get("LM_ICP_TEAM", "default"), "hypotheses": hyp,
                      "device_ms": round(out["device_ms"], 4), "wall_ms": round(out["wall_ms"], 4), "iterations_total": out["iterations_total"],
                      "iters_per_sec_device": round(out["icp_iters_per_sec_device"]), "max_dR_vs_sliced": dR, "max_dt_mm_vs_sliced": dt,
                      "iterations_equal": its == [r["iterations"] for r in ref], "iterations": its}))
    if not sliced:
        for h, d in enumerate(dbg):
            clk = d[25:33]
            if clk[5] > 0 and h < 4:
                print("  hyp %2d evals %2d: workgroup 0, wave-0 cycles per evaluation: exchange %.0f, finish %.0f, transform+queue %.0f, search %.0f, sums %.0f | kernel %.0f cycles, own searches/eval %.1f" % (
                    h, clk[5], clk[7] / clk[5], (clk[0] - clk[7]) / clk[5], clk[1] / clk[5], clk[3] / clk[5], clk[4] / clk[5], clk[2], clk[6] / clk[5]))
                print("           k_icp_knn: slowest workgroup: staging %d, 8-lane trips %d, whole-wave pass %d cycles; %d of %d points went to whole waves" % (d[39], d[40], d[41], d[42], d[38]))
                if os.environ.get("TEAM_MEMBERS") and h in (0, 3):
                    for g in range(16):
                        m = members[h][g]
                        if m[5] > 0:
                            print("      member %2d: per evaluation: exchange %6.0f finish %6.0f move+queue %6.0f search %6.0f sums %6.0f | searches %.1f | kernel %.0f" % (
                                g, m[6] / m[5], (m[0] - m[6]) / m[5], m[1] / m[5], m[2] / m[5], m[3] / m[5], m[4] / m[5], m[7]))
                if os.environ.get("TEAM_ROWS") and h in (0, 3):
                    for it in range(int(clk[5])):
                        r = rows[h][it]
                        print("      eval %2d: classes %s lanes %4d shift %d | cycles: finish %6d (sum+exchange %d solve %d sincos+U %d test %d writes+motion %d) move+queue %6d scatter %5d sweep %6d (thread 0: set-up %d chunks %d merge %d) read %5d sums %6d | motion bound so far %.2f mm" % (
                            it, [int(x) for x in r[:8]], r[8], r[9], r[16], r[20], r[21], r[22], r[23], r[24], r[14], r[10], r[11], r[17], r[18], r[19], r[12], r[15], r[13] * 1e3))
