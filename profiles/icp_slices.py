"""Per-slice shader cycles of the last two ICP evaluations of the long-running hypotheses (icp_big workload)."""
import os, sys
import numpy as np
sys.argv = [sys.argv[0]] + sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(ROOT, "profiles", "icp_big.py")).read().split("for hyp in range")[0])
for hyp in range(n):
    if res[hyp]["iterations"] < 30: continue
    d = ctx.read_debug(hyp, 4).reshape(2, 64, 32)
    for par in range(2):
        tot, sea, pro = d[par, :, 29], d[par, :, 30], d[par, :, 31]
        act = tot > 0
        print("hyp %d parity %d slices %d: total cycles min %.0f median %.0f max %.0f | search min %.0f median %.0f max %.0f | wall(100MHz) median %.0f | corr/slice min %.0f max %.0f" % (
            hyp, par, act.sum(), tot[act].min(), np.median(tot[act]), tot[act].max(), sea[act].min(), np.median(sea[act]), sea[act].max(), np.median(pro[act]), d[par, act, 28].min(), d[par, act, 28].max()))
        print("   totals:", " ".join("%d" % (v / 1000) for v in tot[act]))
