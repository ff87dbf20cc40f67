"""Pins the CPU oracle against the reference's golden vectors — CPU only.

 (1) `test/case1/writeClasses/06_template.yaml` = output of the reference's train_test()
     (linemodLevelup/test.cpp:36-51): reproduced element-for-element.
 (2) Stage hashes and match lists of SURVEY Appendix C.2 on fixture frame 0000 (BGR, T={5,8}).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import linemod_oracle as lo
from helpers import GOLDEN, h16, load_bgr, load_gray, load_u16

EXP = json.load(open(os.path.join(GOLDEN, "expected.json")))


def test_normal_lut_closed_form():
    assert hashlib.sha1(lo.normal_lut().tobytes()).hexdigest() == EXP["normal_lut_sha1"]


def test_similarity_lut_closed_form():
    """Closed form of the active SIMILARITY_LUT (LL.cpp:1121) == numpy response on all 256 values."""
    import ctypes
    lut = np.zeros(256, np.uint8)
    lo.clib().mo_similarity_lut(lut.ctypes.data_as(ctypes.c_void_p))
    v = np.arange(256, dtype=np.uint8).reshape(16, 16)
    resp = lo.response_np(v)
    for ori in range(8):
        lo_t, hi_t = lut[32 * ori:32 * ori + 16], lut[32 * ori + 16:32 * ori + 32]
        got = np.maximum(lo_t[v & 15], hi_t[v >> 4])
        assert np.array_equal(got, resp[ori])
    # first row of the reference table (LL.cpp:1121): {0,4,1,4,0,4,1,4,0,4,1,4,0,4,1,4}
    assert lut[:16].tolist() == [0, 4, 1, 4, 0, 4, 1, 4, 0, 4, 1, 4, 0, 4, 1, 4]


def test_add_template_reproduces_reference_golden(tmp_path):
    rgb, dep, mask = load_bgr("train_rgb.png"), load_u16("train_dep.png"), load_gray("train_mask.png")
    d = lo.OracleDetector()                        # default Detector(): 63 features, T={5,8}
    assert d.addTemplate([rgb, dep], "06_template", mask) == 0
    _, mods, levels, pyr = lo.read_class_yaml(os.path.join(GOLDEN, "writeClasses_06_template.yaml"))
    assert mods == ["ColorGradient", "DepthNormal"] and levels == 2 and len(pyr) == 1
    got = d.class_templates["06_template"][0]
    assert [(t.width, t.height, len(t.features)) for t in got] == [(46, 91, 63), (46, 91, 63), (23, 45, 31), (23, 45, 31)]
    for a, b in zip(got, pyr[0]):
        assert (a.width, a.height, a.pyramid_level) == (b.width, b.height, b.pyramid_level)
        assert np.array_equal(a.features, b.features)
    # YAML round trip through the oracle writer
    d.writeClasses(str(tmp_path / "%s.yaml"))
    _, _, _, pyr2 = lo.read_class_yaml(str(tmp_path / "06_template.yaml"))
    for a, b in zip(pyr2[0], pyr[0]):
        assert (a.width, a.height) == (b.width, b.height) and np.array_equal(a.features, b.features)


def test_train_image_stage_hashes():
    rgb, dep = load_bgr("train_rgb.png"), load_u16("train_dep.png")
    e = EXP["train_bgr"]
    d = lo.OracleDetector()
    assert h16(lo.pyr_down_u8(rgb)) == e["pyrdown"]
    for l, (qc, qn, *_r) in enumerate(d.quantize_pyramid(rgb, dep)):
        assert (h16(qc), int(np.count_nonzero(qc))) == (e["ori"][l], e["ori_nz"][l])
        assert (h16(qn), int(np.count_nonzero(qn))) == (e["nrm"][l], e["nrm_nz"][l])


@pytest.fixture(scope="module")
def frame0000():
    return load_bgr("0000_rgb.png"), load_u16("0000_dep.png")


def test_frame_stage_hashes(frame0000):
    rgb, dep = frame0000
    e = EXP["frame0000_bgr"]
    d = lo.OracleDetector(127, [5, 8])
    assert h16(lo.pyr_down_u8(rgb)) == e["pyrdown"]
    for l, (qc, qn, *_r) in enumerate(d.quantize_pyramid(rgb, dep)):
        T = d.T_at_level[l]
        assert (h16(qc), int(np.count_nonzero(qc))) == (e["ori"][l], e["ori_nz"][l])
        assert (h16(qn), int(np.count_nonzero(qn))) == (e["nrm"][l], e["nrm_nz"][l])
        assert h16(lo.spread_np(qc, T)) == e["spread_ori"][l]
        assert h16(lo.spread_np(qn, T)) == e["spread_nrm"][l]
        n = 8 * qc.size
        lm_c, lm_n = lo.build_linear_memories(qc, T), lo.build_linear_memories(qn, T)
        assert h16(lm_c[:n]) == e["lm_ori"][l] and h16(lm_n[:n]) == e["lm_nrm"][l]
        assert not lm_c[n:].any()
        # C (SSE) == numpy restatement
        rr = lo.response_np(lo.spread_np(qc, T))
        ref = np.concatenate([lo.linearize_np(rr[i], T).reshape(-1) for i in range(8)])
        assert np.array_equal(ref, lm_c[:n])


@pytest.mark.parametrize("bank,nfeat", [("127", 127), ("63", 63)])
def test_match_fixture_frame(frame0000, bank, nfeat, tmp_path):
    """detect_test() inputs (test.cpp:90-128): Detector(127,{5,8}), threshold 75."""
    rgb, dep = frame0000
    e = EXP["match_thr75"][bank]
    d = lo.OracleDetector(nfeat, [5, 8])
    d.readClasses(["06_template"], os.path.join(GOLDEN, "bank" + bank + "_%s.yaml.gz"))
    assert len(d.class_templates["06_template"]) == e["templates"]
    lms, sizes = d.linear_memories(rgb, dep)
    raw = d.match_raw(lms, sizes, 75.0, ["06_template"])
    assert d.last_stats["coarse_candidates"] == e["coarse_candidates"]
    got = sorted([[int(r["x"]), int(r["y"]), float(r["sim"]), int(r["tid"])] for r in raw], key=lambda r: (-r[2], r[3]))
    want = [[x, y, float.fromhex(s), t] for x, y, s, t in e["pre_unique"]]
    assert got == want
    # canonical merge: one entry per distinct (x,y,sim,class)
    final = lo.canonical_sort_unique(raw)
    assert len(final) == len({(x, y, s) for x, y, s, _ in want})
    # threaded variant returns the same multiset
    raw4 = d.match_raw(lms, sizes, 75.0, ["06_template"], nthreads=4)
    assert sorted(raw4.tolist()) == sorted(raw.tolist())
    # GT bbox origin (331,130) (test.cpp:86): top match within a few px — sanity only
    assert abs(got[0][0] - 331) <= 4 and abs(got[0][1] - 130) <= 4
