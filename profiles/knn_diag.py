"""k_icp_knn per-point diagnostics (cov pad slot): passes, ring candidates, cycles."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(ROOT, "profiles", "icp_big.py")).read().split("for hyp in range")[0])
for hyp in range(min(n, 4)):
    c = ctx.read_debug(hyp, 5).reshape(-1, 12)
    if not len(c): continue
    x = c[:, 11]
    A, Bp = x[x > 0], -x[x < 0]
    print("hyp %d n_tgt %d: phase A points %d cycles median %.0f max %.0f | phase B points %d cycles median %.0f max %.0f sum %.0f" % (
        hyp, len(c), len(A), np.median(A) if len(A) else 0, A.max() if len(A) else 0, len(Bp), np.median(Bp) if len(Bp) else 0, Bp.max() if len(Bp) else 0, Bp.sum()))
