"""Cases on which the CPU oracle (oracle/match_oracle.c) and the GPU path are pinned to the
reference's own match lines (oracle/_ref, see oracle/Makefile).  Shared by
tests/golden/make_ref_fixtures.py (writes tests/golden/ref_expected.json from oracle/_ref),
tests/test_ref_pin.py (CPU) and tests/test_gpu_parity.py (GPU)."""
import hashlib
import os

import numpy as np

import linemod_oracle as lo
import synth
from helpers import GOLDEN, load_bgr, load_u16

FIXTURE_THRESHOLDS = (75.0, 55.0)


def sha(a) -> str:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_case(frame: str, bank: str):
    """detect_test() inputs of linemodLevelup/test.cpp:90-128: frame 0000 (or its half-occluded
    variant, test.cpp:95-96) read as BGR, Detector(nfeat, {5, 8}), banks 63 / 127 / 600."""
    nfeat = 63 if bank == "63" else 127
    rgb, dep = load_bgr("0000_rgb%s.png" % frame), load_u16("0000_dep%s.png" % frame)
    od = lo.OracleDetector(nfeat, [5, 8])
    od.readClasses(["06_template"], os.path.join(GOLDEN, "bank" + bank + "_%s.yaml.gz"))
    return dict(name="fixture%s_bank%s" % (frame, bank), rgb=rgb, dep=dep, T=[5, 8], nfeat=nfeat, od=od,
                banks={"06_template": lo.pack_bank(od.class_templates["06_template"], 2)},
                requests=[["06_template"]], thresholds=FIXTURE_THRESHOLDS)


SYNTH = [  # W, H, T, nfeat per level, planted templates, threshold, frame seed
    (640, 480, [4, 8], (150, 75), 160, 75.0, 11),      # Detector(150,[4,8]) of the driver script
    (640, 480, [4, 8], (63, 31), 120, 70.0, 12),       # < 64 features: the reference's 8-bit path
    (320, 240, [4, 8], (64, 32), 80, 65.0, 13),
    (640, 480, [4, 4, 8], (64, 32, 16), 80, 70.0, 14),  # three pyramid levels
    (640, 480, [8], (75,), 100, 75.0, 15),             # single level: no refinement
    (1280, 960, [4, 8], (150, 75), 60, 75.0, 16),
]


def synth_case(i: int):
    W, H, T, nfeat, n, thr, seed = SYNTH[i]
    rgb, dep = synth.make_frame(seed, W, H, 40 if W <= 640 else 80)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    planted = synth.make_planted_bank(seed + 10, n, [(p[0], p[1]) for p in pyr], T, nfeat)
    random = synth.make_random_bank(seed + 20, n // 2, W, H, nfeat)
    L = len(T)
    banks = {"planted": lo.PackedBank(n, L, *planted), "random": lo.PackedBank(n // 2, L, *random)}
    return dict(name="synth%d_%dx%d_T%s_f%d" % (i, W, H, "".join(map(str, T)), nfeat[0]), rgb=rgb, dep=dep, T=T,
                nfeat=nfeat[0], od=od, banks=banks, pyr=pyr,
                requests=[["planted", "random"], ["random", "nope", "planted"], []], thresholds=(thr,))


def all_cases():
    for frame in ("", "_half"):
        for bank in ("63", "127", "600"):
            yield fixture_case(frame, bank)
    for i in range(len(SYNTH)):
        yield synth_case(i)


def quantized_of(case):
    pyr = case.get("pyr") or case["od"].quantize_pyramid(case["rgb"], case["dep"])
    case["pyr"] = pyr
    return [(p[0], p[1]) for p in pyr]


def oracle_run(case, thr, req):
    """Pre-unique list of the oracle, `cls` = index into sorted(banks) like ll_ref.match."""
    od, T = case["od"], case["T"]
    if "lms" not in case:
        q = quantized_of(case)
        case["lms"] = [[lo.build_linear_memories(q[l][m], T[l]) for m in range(2)] for l in range(len(T))]
        case["sizes"] = [(q[l][0].shape[1], q[l][0].shape[0]) for l in range(len(T))]
    names = sorted(case["banks"])
    order = list(req) if req else names
    out, cand = [], 0
    for cid in order:
        if cid not in case["banks"]:
            continue
        m, st = lo.match_bank_c(case["banks"][cid], case["lms"], case["sizes"], T, thr)
        m["cls"] = names.index(cid)
        cand += st["coarse_candidates"]
        out.append(m)
    return (np.concatenate(out) if out else np.zeros(0, lo.MATCH_DTYPE)), cand


def record_key(thr, req):
    return "thr%g|%s" % (thr, ",".join(req))
