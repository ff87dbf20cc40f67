"""GPU parity tests (`-m gpu`): the HIP path, called through the C ABI, against the CPU oracle on
the same inputs — bit-exact for every integer/byte stage and for (x, y, similarity, template_id);
ICP within 1e-4 on R and t (metres) as BASELINE.json's north_star states."""
import os

import numpy as np
import pytest

import linemod_oracle as lo
import synth
from helpers import GOLDEN, load_bgr, load_gray, load_u16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lm():
    import __graft_entry__ as g
    import linemodLevelup_pybind as mod
    if not os.path.exists(mod.library_path()):
        g.build()
    lib = mod.load_library()
    assert lib.lm_device_count() >= 1, "GPU tests need a visible MI355X (no CPU fallback)"
    return mod


def oracle_matches(od, rgb, dep, bank_arrays, T, thr, cls=0):
    feat, offs, wh = bank_arrays
    lms, sizes = od.linear_memories(rgb, dep)
    P = (len(offs) - 1) // (2 * len(T))
    raw, st = lo.match_bank_c(lo.PackedBank(P, len(T), feat, offs, wh), lms, sizes, T, thr)
    raw["cls"] = cls
    return raw, st


def same_records(got, want):
    """got: product MATCH_DTYPE, want: oracle MATCH_DTYPE; exact comparison field by field."""
    assert len(got) == len(want), (len(got), len(want))
    for g, w in (("x", "x"), ("y", "y"), ("similarity", "sim"), ("class_index", "cls"), ("template_id", "tid")):
        assert np.array_equal(got[g], want[w]), g


def as_multiset(rec, names):
    return sorted(zip(*[rec[n].tolist() for n in names]))


# every kernel path of the matcher (Detector.setPaths: refinement, coarse pass); results may never depend on it
PATHS = [("bits", "bits"), ("bits", "bytes"), ("tiles", "bytes"), ("single", "bytes"),
         ("bits", "bits", False)]        # ... and the bit planes packed from byte linear memories instead of written by the front end itself


def detector_on(lm, paths, *args, **kw):
    det = lm.Detector(*args, **kw)
    det.setPaths(*paths)
    return det


def expect_paths(det, paths, tiles_possible=True, levels=2):
    """After a match: the kernels in use are the ones the case asked for — a test must not compare a path with itself."""
    refine, coarse = paths[:2]
    if levels < 2:
        want = ("single", "bytes")
    elif refine == "tiles" and not tiles_possible:
        want = ("single", "bytes")
    else:
        want = (refine, coarse)
    assert det.getPaths() == want, (det.getPaths(), want)
    assert det.refinesOnBitPlanes() == (want[0] == "bits")


# ---------------------------------------------------------------------------------------------
# front end: quantised maps and linear memories, byte for byte
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["fixture_T58", "synth_T48", "synth_small_T48", "synth_3level", "synth_1280", "synth_100x96_T22", "synth_50x48_T2"])
def test_frontend_stages_bit_exact(lm, case):
    if case == "fixture_T58":
        rgb, dep, T, nf = load_bgr("0000_rgb.png"), load_u16("0000_dep.png"), [5, 8], 127
    elif case == "synth_T48":
        (rgb, dep), T, nf = synth.make_frame(0), [4, 8], 150
    elif case == "synth_small_T48":
        (rgb, dep), T, nf = synth.make_frame(5, 320, 240, 12), [4, 8], 64
    elif case == "synth_3level":
        (rgb, dep), T, nf = synth.make_frame(6, 640, 480), [4, 4, 8], 64
    elif case == "synth_100x96_T22":                     # level 1 is 50 wide: rows that are not dword-aligned, tiles hanging over the right edge (the four-pixel stores of the vote / the median fall back to bytes)
        (rgb, dep), T, nf = synth.make_frame(8, 100, 96, 10), [2, 2], 64
    elif case == "synth_50x48_T2":
        (rgb, dep), T, nf = synth.make_frame(8, 50, 48, 6), [2], 64
    else:
        (rgb, dep), T, nf = synth.make_frame(7, 1280, 960, 60), [4, 8], 150
    od = lo.OracleDetector(nf, T)
    pyr = od.quantize_pyramid(rgb, dep)
    det = lm.Detector(nf, T, device=0)
    det.addClassPacked("e", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    det.setFrame([rgb, dep])
    det.matchResident(75.0, ["e"])
    for l, (qc, qn, *_r) in enumerate(pyr):
        assert np.array_equal(det.readStage(l, 0).reshape(qc.shape), qc), "orientations level %d" % l
        assert np.array_equal(det.readStage(l, 1).reshape(qn.shape), qn), "normals level %d" % l
        n = 8 * qc.size
        assert np.array_equal(det.readStage(l, 2), lo.build_linear_memories(qc, T[l])[:n]), "LM colour level %d" % l
        assert np.array_equal(det.readStage(l, 3), lo.build_linear_memories(qn, T[l])[:n]), "LM normal level %d" % l


def test_frontend_masks(lm):
    rgb, dep = synth.make_frame(2, 320, 240, 12)
    T = [4, 8]
    rng = np.random.default_rng(0)
    m0 = (rng.uniform(0, 1, dep.shape) < 0.7).astype(np.uint8) * 255
    m1 = np.zeros(dep.shape, np.uint8)
    m1[40:200, 30:290] = 1
    od = lo.OracleDetector(64, T)
    pyr = od.quantize_pyramid(rgb, dep)
    det = lm.Detector(64, T, device=0)
    det.addClassPacked("e", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    det.setFrame([rgb, dep], [m0, m1])
    det.matchResident(75.0, [])
    a, b = m0, m1
    for l, (qc, qn, *_r) in enumerate(pyr):
        if l > 0:
            a, b = lo.nn_down2(a), lo.nn_down2(b)
        n = 8 * qc.size
        assert np.array_equal(det.readStage(l, 2), lo.build_linear_memories(np.where(a > 0, qc, 0).astype(np.uint8), T[l])[:n])
        assert np.array_equal(det.readStage(l, 3), lo.build_linear_memories(np.where(b > 0, qn, 0).astype(np.uint8), T[l])[:n])


def _strip_records(lm_flat, T, Wd, Hd):
    """numpy statement of the strip records (match.hip): [label][phase][strip][row] uint64, cell c of [16 s, 16 s + 32) of the row at bits
    2c (response is 1) and 2c + 1 (response is 4), from a flat linear memory [8][T*T][Hd*Wd]."""
    NS = (Wd + 15) // 16
    planes = lm_flat[:8 * T * T * Wd * Hd].reshape(8, T * T, Hd, Wd)
    pad = np.zeros((8, T * T, Hd, NS * 16 + 32), np.uint8)
    pad[..., :Wd] = planes
    w1 = np.uint64(1) << (2 * np.arange(32, dtype=np.uint64))
    out = np.zeros((8, T * T, NS, Hd), np.uint64)
    for s in range(NS):
        seg = pad[..., 16 * s:16 * s + 32]
        out[:, :, s, :] = ((seg == 1).astype(np.uint64) * w1).sum(axis=-1) + ((seg == 4).astype(np.uint64) * (w1 << np.uint64(1))).sum(axis=-1)
    return out


def _pair_stream(lm_colour, lm_normal, T, Wd, Hd, npairs):
    """numpy statement of the top level's pair stream: {is-1 dword, is-4 dword} per 32 bytes of the two modality blocks (zero tails included)."""
    block = npairs * 16
    flat = np.zeros(2 * block, np.uint8)
    n = 8 * T * T * Wd * Hd
    flat[:n] = lm_colour[:n]; flat[block:block + n] = lm_normal[:n]
    g = flat.reshape(npairs, 32)
    w = np.uint32(1) << np.arange(32, dtype=np.uint32)
    out = np.zeros((npairs, 2), np.uint32)
    out[:, 0] = ((g == 1).astype(np.uint32) * w).sum(axis=1); out[:, 1] = ((g == 4).astype(np.uint32) * w).sum(axis=1)
    return out


@pytest.mark.parametrize("case", ["T48", "T58", "3level", "1280", "small_T24", "masked", "masked_T58"])
def test_bit_planes_equal_the_packed_linear_memories(lm, case):
    """The encodings the bit-plane kernels read — strip records of every level below the top, pair stream of the top level —
    bit for bit against a numpy packing of the ORACLE's linear memories, as written directly by the front end (k_fe_bits: strip records from
    LDS cell words; the pair stream as whole dwords, or OR-ed together from shifted ballots) and as packed from the byte planes (k_pack_bits /
    k_pack_top)."""
    W, H, T = {"T48": (640, 480, [4, 8]), "T58": (640, 480, [5, 8]), "3level": (640, 480, [4, 4, 8]), "1280": (1280, 960, [4, 8]),
               "small_T24": (320, 240, [2, 4]), "masked": (640, 480, [4, 8]), "masked_T58": (640, 480, [5, 8])}[case]
    rgb, dep = synth.make_frame(61, W, H, 40 if W <= 640 else 80)
    masks = []
    if case.startswith("masked"):
        yy, xx = np.mgrid[0:H, 0:W]
        masks = [((xx // 37 + yy // 29) % 3 != 0).astype(np.uint8) * 255, ((xx // 23 + yy // 41) % 4 != 0).astype(np.uint8) * 255]
    nfeat = tuple(64 >> l for l in range(len(T)))
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    bank = synth.make_planted_bank(62, 12, [(p[0], p[1]) for p in pyr], T, nfeat)
    lms, mk = [], list(masks)
    for l, p in enumerate(pyr):
        if masks and l > 0:
            mk = [lo.nn_down2(m) for m in mk]                 # the masks follow the pyramid by nearest neighbour (LL.cpp:573-578, 874-879)
        q = [np.where(mk[m] > 0, p[m], 0).astype(np.uint8) if masks else p[m] for m in range(2)]
        lms.append([lo.build_linear_memories(q[0], T[l]), lo.build_linear_memories(q[1], T[l])])
    L = len(T)
    seen = []
    for direct in (2, 10, 6, False):                          # the front end's own writers (2: kept readable — pixel tiles where the geometry allows; 10: ... whole dwords per wave; 6: ... the OR-ing writer), packed from bytes
        det = lm.Detector(nfeat[0], T, device=0)
        det.setPaths("bits", "bits", direct)
        det.addClassPacked("o", *bank)
        det.setFrame([rgb, dep], masks)
        det.matchResident(70.0, ["o"])
        assert det.getPaths() == ("bits", "bits")
        got = []
        for l in range(L - 1):
            Wd, Hd = (W >> l) // T[l], (H >> l) // T[l]
            NS = (Wd + 15) // 16
            rec = det.readStage(l, 4).view(np.uint64).reshape(2, 8, T[l] * T[l], NS, Hd)
            for m in range(2):
                assert np.array_equal(rec[m], _strip_records(lms[l][m], T[l], Wd, Hd)), (case, direct, "strip records", l, m)
            got.append(rec.tobytes())
        Wd, Hd = (W >> (L - 1)) // T[-1], (H >> (L - 1)) // T[-1]
        ps = det.readStage(L - 1, 5).view(np.uint32).reshape(-1, 2)
        assert np.array_equal(ps, _pair_stream(lms[L - 1][0], lms[L - 1][1], T[-1], Wd, Hd, len(ps))), (case, direct, "pair stream")
        got.append(ps.tobytes())
        seen.append(got)
        # and the stream is usable again: the next match clears what this one kept
        if not masks:
            raw, _ = oracle_matches(od, rgb, dep, bank, T, 70.0)
            same_records(det.matchArray([rgb, dep], 70.0, ["o"]), lo.canonical_sort_unique(raw))
    assert seen[0] == seen[1] == seen[2] == seen[3]


def test_precondition_errors_like_cv_assert(lm):
    det = lm.Detector(63, [5, 8], device=0)
    rgb, dep = synth.make_frame(1, 320, 240, 4)                 # 160x120 at level 1: 120 % 8 == 0 but 320 % 5 == 0, 240%5==0 ok
    bad_rgb, bad_dep = rgb[:, :318].copy(), dep[:, :318].copy()    # 318 % 5 != 0
    with pytest.raises(RuntimeError, match=r"% T == 0|% 16 == 0"):
        det.match([np.ascontiguousarray(bad_rgb), np.ascontiguousarray(bad_dep)], 75.0, [])
    with pytest.raises(RuntimeError):
        det.match([rgb, dep.astype(np.float32)], 75.0, [])
    with pytest.raises(RuntimeError):
        det.match([rgb], 75.0, [])


# ---------------------------------------------------------------------------------------------
# match: fixture banks (reference detect_test inputs) and synthetic banks
# ---------------------------------------------------------------------------------------------
def test_gpu_equals_the_reference_lines(lm):
    """The GPU path against oracle/_ref = the reference's OWN match code (LL.cpp:1022-1658, 1694-1941 compiled unmodified against a
    cv::Mat buffer shim, oracle/Makefile) on every reference fixture: frame 0000 and its half-occluded variant (test.cpp:95-96)
    x banks 63 / 127 / 600 (test.cpp:111-123), thresholds 75 (test.cpp:128) and 55.  Linear memories byte for byte, the
    pre-unique match list as a multiset, and the distinct (x, y, similarity) of Detector::match's own output.  Where the prebuilt
    library did not travel, the committed digest tests/golden/ref_expected.json (written from it) is used instead."""
    import json
    import ll_ref
    import ref_cases as rc
    exp = json.load(open(os.path.join(GOLDEN, "ref_expected.json")))
    for frame in ("", "_half"):
        for bank in ("63", "127", "600"):
            case = rc.fixture_case(frame, bank)
            q, T = rc.quantized_of(case), case["T"]
            det = lm.Detector(case["nfeat"], T, device=0)
            det.readClasses(["06_template"], os.path.join(GOLDEN, "bank" + bank + "_%s.yaml.gz"))
            det.setFrame([case["rgb"], case["dep"]])
            det.matchResident(75.0, ["06_template"])
            for l in range(2):
                for m in range(2):
                    lm_gpu = det.readStage(l, 2 + m)[:8 * q[l][m].size]
                    assert rc.sha(lm_gpu) == exp[case["name"]]["lm"][l][m]
                    if ll_ref.available():
                        assert np.array_equal(lm_gpu, ll_ref.build_linear_memories(q[l][m], T[l]))
            for thr in rc.FIXTURE_THRESHOLDS:
                e = exp[case["name"]]["match"][rc.record_key(thr, ["06_template"])]
                pre = det.matchResident(thr, ["06_template"], sort_unique=False)
                fin = det.matchResident(thr, ["06_template"])
                assert len(pre) == e["pre_unique_n"]
                distinct = set(zip(fin["x"].tolist(), fin["y"].tolist(), fin["similarity"].tolist()))
                assert len(distinct) == e["final_distinct_n"]
                if ll_ref.available():
                    rpre = ll_ref.match(q, T, case["banks"], thr, ["06_template"], pre_unique=True)
                    assert as_multiset(pre, ["x", "y", "similarity", "template_id"]) == as_multiset(rpre, ["x", "y", "sim", "tid"])
                    rfin = ll_ref.match(q, T, case["banks"], thr, ["06_template"])
                    assert distinct == set(zip(rfin["x"].tolist(), rfin["y"].tolist(), rfin["sim"].tolist()))
                    if len(rfin):
                        assert (int(fin[0]["x"]), int(fin[0]["y"]), float(fin[0]["similarity"])) == (int(rfin[0]["x"]), int(rfin[0]["y"]), float(rfin[0]["sim"]))
                    # reference-order mode: Detector::match's own list, entry by entry — the permutation libstdc++'s std::sort
                    # leaves and the duplicates std::unique keeps included (560 entries on bank 63 at threshold 55, not 366)
                    det.setReferenceOrder(True)
                    exact = det.matchResident(thr, ["06_template"])
                    det.setReferenceOrder(False)
                    assert len(exact) == e["final_n"] == len(rfin)
                    for a, b in (("x", "x"), ("y", "y"), ("similarity", "sim"), ("template_id", "tid")):
                        assert np.array_equal(exact[a], rfin[b]), (case["name"], thr, a)
    # two classes in the caller's order, an unknown class in between, 8-bit and 16-bit paths, tiles on: still entry by entry
    if ll_ref.available():
        for i in (0, 1):
            case = rc.synth_case(i)
            q, T = rc.quantized_of(case), case["T"]
            det = lm.Detector(case["nfeat"], T, device=0)
            for name, b in case["banks"].items():
                det.addClassPacked(name, b.feat, b.tmpl_off, b.tmpl_wh)
            det.setReferenceOrder(True)
            names = sorted(case["banks"])
            for req in case["requests"]:
                thr = case["thresholds"][0]
                got = det.matchArray([case["rgb"], case["dep"]], thr, req)
                rfin = ll_ref.match(q, T, case["banks"], thr, req)
                order = list(req) if req else names
                assert len(got) == len(rfin) > 0
                assert [order[k] for k in got["class_index"].tolist()] == [names[k] for k in rfin["cls"].tolist()]
                for a, b in (("x", "x"), ("y", "y"), ("similarity", "sim"), ("template_id", "tid")):
                    assert np.array_equal(got[a], rfin[b]), (case["name"], req, a)


@pytest.mark.parametrize("paths", PATHS, ids=["-".join(map(str, p)) for p in PATHS])
@pytest.mark.parametrize("bank,nfeat", [("127", 127), ("63", 63)])
def test_match_fixture_banks(lm, bank, nfeat, paths):
    rgb, dep = load_bgr("0000_rgb.png"), load_u16("0000_dep.png")
    fmt = os.path.join(GOLDEN, "bank" + bank + "_%s.yaml.gz")
    od = lo.OracleDetector(nfeat, [5, 8])
    od.readClasses(["06_template"], fmt)
    det = detector_on(lm, paths, nfeat, [5, 8], device=0)
    det.readClasses(["06_template"], fmt)
    assert det.numTemplates("06_template") == 89 and det.classIds() == ["06_template"]
    # bank round trip: what the product parsed equals what the oracle parsed
    for tid in (0, 34, 88):
        for a, b in zip(det.getTemplates("06_template", tid), od.class_templates["06_template"][tid]):
            assert (a.width, a.height, a.pyramid_level) == (b.width, b.height, b.pyramid_level)
            assert np.array_equal(a.features, b.features)
    for thr in (75.0, 60.0):
        got = det.matchArray([rgb, dep], thr, ["06_template"])
        lms, sizes = od.linear_memories(rgb, dep)
        raw = od.match_raw(lms, sizes, thr, ["06_template"])
        same_records(got, lo.canonical_sort_unique(raw))
        expect_paths(det, paths)
        tm = det.lastTimings()
        assert tm["coarse_candidates"] == od.last_stats["coarse_candidates"]
        assert tm["matches_pre_unique"] == len(raw)
        # pre-unique multiset
        det.setFrame([rgb, dep])
        pre = det.matchResident(thr, ["06_template"], sort_unique=False)
        assert as_multiset(pre, ["x", "y", "similarity", "template_id"]) == as_multiset(raw, ["x", "y", "sim", "tid"])
    ms = det.match([rgb, dep], 75.0, ["06_template"], masks=[])
    assert (ms[0].x, ms[0].y, ms[0].template_id, ms[0].class_id) == (332, 127, 34, "06_template")


@pytest.mark.parametrize("W,H,T,nfeat,n,thr", [
    (640, 480, [4, 8], (150, 75), 300, 75.0),       # Detector(150,[4,8]) of the driver script
    (640, 480, [4, 8], (63, 31), 200, 70.0),        # < 64 features: the reference's 8-bit path
    (320, 240, [4, 8], (64, 32), 100, 65.0),
    (640, 480, [4, 4, 8], (64, 32, 16), 120, 70.0), # three pyramid levels
    (640, 480, [8], (75,), 150, 75.0),              # single level: no refinement
    (1280, 960, [4, 8], (150, 75), 120, 75.0),
])
def test_match_planted_and_random_banks(lm, W, H, T, nfeat, n, thr):
    rgb, dep = synth.make_frame(11, W, H, 40 if W <= 640 else 80)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    planted = synth.make_planted_bank(21, n, [(p[0], p[1]) for p in pyr], T, nfeat)
    random = synth.make_random_bank(22, n // 2, W, H, nfeat)
    ra, sa = oracle_matches(od, rgb, dep, planted, T, thr, 0)
    rb, sb = oracle_matches(od, rgb, dep, random, T, thr, 1)
    assert sa["coarse_candidates"] > n, "planted bank must exercise the refinement"
    want = lo.canonical_sort_unique(np.concatenate([ra, rb]))
    for paths in PATHS[1:]:                                       # every other path (the default, bit planes everywhere, follows with the further checks)
        dp = detector_on(lm, paths, nfeat[0], T, device=0)
        dp.addClassPacked("planted", *planted)
        dp.addClassPacked("random", *random)
        same_records(dp.matchArray([rgb, dep], thr, ["planted", "random"]), want)
        expect_paths(dp, paths, tiles_possible=len(T) == 2, levels=len(T))
        tmp_ = dp.lastTimings()
        assert tmp_["coarse_candidates"] == sa["coarse_candidates"] + sb["coarse_candidates"]
        assert tmp_["local_evals"] == sa["local_evals"] + sb["local_evals"]
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("planted", *planted)
    det.addClassPacked("random", *random)
    got = det.matchArray([rgb, dep], thr, ["planted", "random"])
    same_records(got, want)
    expect_paths(det, PATHS[0], levels=len(T))                   # three levels too: every level below the top on bit planes
    assert det.lastTimings()["coarse_candidates"] == sa["coarse_candidates"] + sb["coarse_candidates"]
    assert det.lastTimings()["local_evals"] == sa["local_evals"] + sb["local_evals"]
    # class order given by the caller; unknown classes are skipped; empty list = sorted class order
    got2 = det.matchArray([rgb, dep], thr, ["random", "nope", "planted"])
    rb2, ra2 = rb.copy(), ra.copy()
    rb2["cls"] = 0; ra2["cls"] = 2
    same_records(got2, lo.canonical_sort_unique(np.concatenate([rb2, ra2])))
    same_records(det.matchArray([rgb, dep], thr, []), want)     # sorted: planted < random
    # idempotence, and the HBM frame-slot path gives the same result
    same_records(det.matchArray([rgb, dep], thr, ["planted", "random"]), want)
    det.storeFrame(1, [rgb, dep])
    det.storeFrame(0, synth.make_frame(99, W, H, 5))
    det.selectFrame(0)
    det.matchResident(thr, ["planted", "random"])
    det.selectFrame(1)
    same_records(det.matchResident(thr, ["planted", "random"]), want)


def test_match_edge_cases(lm):
    W, H, T = 320, 240, [4, 8]
    rgb, dep = synth.make_frame(4, W, H, 14)
    od = lo.OracleDetector(32, T)
    pyr = od.quantize_pyramid(rgb, dep)
    det = lm.Detector(32, T, device=0)
    # empty detector / empty class / unknown class
    assert len(det.matchArray([rgb, dep], 75.0, [])) == 0
    det.addClassPacked("empty", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    assert len(det.matchArray([rgb, dep], 75.0, ["empty", "unknown"])) == 0
    # hand-made pyramids: template larger than the image (template_positions <= 0), negative and
    # out-of-image feature coordinates (discarded by LL.cpp:1330 / :1394), features at x==width
    qc1 = pyr[1][0]
    ys, xs = np.nonzero(qc1)
    lab = np.log2(qc1[ys, xs]).astype(np.int32)
    def entry(fe, w, h):
        return np.asarray(fe, np.int32).reshape(-1, 3), (w, h)
    base0 = [[int(x) * 2, int(y) * 2, int(l)] for x, y, l in zip(xs[:40] % 60, ys[:40] % 60, lab[:40])]
    base1 = [[int(x), int(y), int(l)] for x, y, l in zip(xs[:20] % 30, ys[:20] % 30, lab[:20])]
    pyrs = [
        [entry(base0, 120, 120), entry(base0, 120, 120), entry(base1, 60, 60), entry(base1, 60, 60)],
        [entry(base0, 400, 300), entry(base0, 400, 300), entry(base1, 200, 150), entry(base1, 200, 150)],      # too large
        [entry(base0 + [[-3, 5, 1], [5000, 2, 2]], 120, 120), entry(base0, 120, 120),
         entry(base1 + [[-1, -1, 0], [161, 3, 3]], 60, 60), entry(base1 + [[60, 60, 7]], 60, 60)],
        [entry([[0, 0, 0]], 0, 0), entry([[0, 0, 1]], 0, 0), entry([[0, 0, 0]], 0, 0), entry([[0, 0, 1]], 0, 0)],   # 1-feature, size 0
    ]
    feats, offs, whs = [], [0], []
    for p in pyrs:
        for f, wh in p:
            feats.append(f); offs.append(offs[-1] + len(f)); whs.append(wh)
    bank = (np.concatenate(feats), np.asarray(offs, np.int32), np.asarray(whs, np.int32))
    det.addClassPacked("hand", *bank)
    for thr in (10.0, 50.0, 100.0, 0.0):
        want, _ = oracle_matches(od, rgb, dep, bank, T, thr, 0)
        for paths in PATHS:                                     # (bit planes: the oversized and out-of-frame templates take k_local's per-candidate path in a second launch)
            det.setPaths(*paths)
            same_records(det.matchArray([rgb, dep], thr, ["hand"]), lo.canonical_sort_unique(want))
            expect_paths(det, paths)
    det.setPaths("bits", "bits")
    # candidate-buffer growth path: threshold 0 on a bigger bank overflows the initial capacity? (count only)
    assert det.lastTimings()["matches_pre_unique"] == len(want)
    # invalid banks are rejected like the reference's CV_Asserts
    with pytest.raises(RuntimeError, match="8191"):
        big = np.zeros((8192, 3), np.int32)
        det.addClassPacked("big", np.concatenate([big, big, big, big]), np.arange(5, dtype=np.int32) * 8192,
                           np.full((4, 2), 10, np.int32))
    with pytest.raises(RuntimeError):
        det.addClassPacked("hand", *bank)               # class already present


def test_refinement_paths_are_exact(lm):
    """Four kernel paths serve Detector.match — bit planes for both passes (k_coarse_bits + k_local_bits, the default), bit-plane refinement
    behind the byte coarse pass, the byte strip planes with tiles (candidates of a template whose coarse cells are neighbours, planned
    by k_coarse, summed once per tile in k_local) and every candidate on its own (round 1).  Each must give the records of the oracle
    and of the others — pre-unique multiset, evaluation count and algorithmic bytes included — on a planted bank (clusters of
    neighbouring candidates), a random one (isolated candidates), with templates planted against the frame border (clamped windows),
    with oversized templates on the slow path, at low thresholds, and with frames in flight; `getPaths` proves which kernels ran."""
    W, H, T, nfeat = 640, 480, [4, 8], (150, 75)
    rgb, dep = synth.make_frame(5, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    banks = {"planted": synth.make_planted_bank(17, 300, [(p[0], p[1]) for p in pyr], T, nfeat), "random": synth.make_random_bank(18, 200, W, H, nfeat)}
    dets = {}
    for paths in PATHS:
        d = detector_on(lm, paths, nfeat[0], T, device=0)
        for c, b in banks.items():
            d.addClassPacked(c, *b)
        d.setFrame([rgb, dep])
        dets[paths] = d
    plain = dets[("single", "bytes")]
    names = ["x", "y", "similarity", "class_index", "template_id"]
    for thr, ids in ((75.0, ["planted"]), (60.0, ["random", "planted"]), (88.0, []), (45.0, ["planted"])):
        a = plain.matchResident(thr, ids)
        ta = plain.lastTimings()
        ra = plain.matchResident(thr, ids, sort_unique=False)
        assert len(a) > 0
        for paths in PATHS:
            if paths == ("single", "bytes"):
                continue
            d = dets[paths]
            b = d.matchResident(thr, ids)
            expect_paths(d, paths)
            assert a.tobytes() == b.tobytes(), (paths, thr, ids, len(a), len(b))
            tb = d.lastTimings()
            assert ta["coarse_candidates"] == tb["coarse_candidates"], paths
            assert ta["local_evals"] == tb["local_evals"] and ta["matches_pre_unique"] == tb["matches_pre_unique"] and ta["local_bytes"] == tb["local_bytes"], paths
            rb = d.matchResident(thr, ids, sort_unique=False)
            assert as_multiset(ra, names) == as_multiset(rb, names), paths
        if thr >= 60.0:                                           # ... and the oracle itself
            order = ids if ids else sorted(banks)
            raws = []
            for ci, c in enumerate(order):
                raw, _ = oracle_matches(od, rgb, dep, banks[c], T, thr, ci)
                raws.append(raw)
            same_records(a, lo.canonical_sort_unique(np.concatenate(raws)))
    want75 = plain.matchResident(75.0, ["planted"]).tobytes()
    for paths in PATHS:                                         # pipelined, slots reused
        d = dets[paths]
        for k in range(7):
            d.submit(75.0, ["planted"])
            if k >= 2:
                assert d.collect().tobytes() == want75, paths
        d.collect(); d.collect()
    # other geometries: T = {2, 4} and {8, 16} (coarse cells again 4 fine cells apart), a small and a non-square frame; the reference's
    # default T = {5, 8} (Detector(), LL.cpp:1663-1692: every fixture bank): the windows of neighbouring coarse cells lie 16 / 5 level-0
    # cells apart - 3, 3, 3, 3, 4, ... - and the tiles carry those steps; T = {6, 8} (steps 2 and 3), {3, 4}; T = {4, 16} (8 cells apart) is
    # not tiled.  Every path == oracle, statistics included.
    for (W5, H5, T5, nf5, thr5, seed5) in ((320, 240, [2, 4], (64, 32), 65.0, 29), (640, 480, [8, 16], (64, 32), 60.0, 29), (384, 272, [4, 8], (96, 48), 65.0, 29),
                                           (640, 480, [5, 8], (63, 31), 70.0, 19), (640, 480, [5, 8], (127, 63), 60.0, 20), (768, 480, [6, 8], (64, 32), 65.0, 21),
                                           (768, 480, [3, 4], (64, 32), 65.0, 22), (640, 480, [4, 16], (64, 32), 60.0, 23)):
        rgb5, dep5 = synth.make_frame(5 if T5[0] in (3, 5, 6) or T5 == [4, 16] else 23, W5, H5, 40 if W5 >= 640 else 24)
        od5 = lo.OracleDetector(nf5[0], T5)
        p5 = od5.quantize_pyramid(rgb5, dep5)
        b5 = synth.make_planted_bank(seed5, 160, [(p[0], p[1]) for p in p5], T5, nf5)
        raw, st5 = oracle_matches(od5, rgb5, dep5, b5, T5, thr5)
        want5 = lo.canonical_sort_unique(raw)
        assert len(want5) > 0 and st5["coarse_candidates"] > 100
        stats = []
        for paths in PATHS:
            d5 = detector_on(lm, paths, nf5[0], T5, device=0)
            d5.addClassPacked("o", *b5)
            same_records(d5.matchArray([rgb5, dep5], thr5, ["o"]), want5)
            expect_paths(d5, paths, tiles_possible=T5 != [4, 16])
            tm5 = d5.lastTimings()
            assert tm5["coarse_candidates"] == st5["coarse_candidates"]
            stats.append((tm5["local_evals"], tm5["local_bytes"], tm5["matches_pre_unique"]))
        assert len(set(stats)) == 1 and stats[0][0] == st5["local_evals"], (T5, stats)


def test_feature_count_boundaries_and_mixed_banks(lm):
    """The bit-sliced counters of the bit-plane kernels hold 511 features per template entry (both modalities of a pyramid level) in their
    small instantiation; larger entries — the reference allows 8191 per modality (LL.cpp:1291, 1816) — take the 14-bit one.  Banks at the
    boundary (511 / 512 features per level-0 entry), a bank mixing small and large templates, one with 2150-feature entries, and a
    three-level pyramid with a 600-feature middle level: every kernel path equals the oracle, statistics included."""
    W, H = 640, 480
    rgb, dep = synth.make_frame(31, W, H, 40)
    cases = [
        ("511", [4, 8], [((256, 255), (128, 127))], 80, 72.0),
        ("512", [4, 8], [((256, 256), (128, 128))], 80, 72.0),
        ("mixed", [4, 8], [((150, 150), (75, 75)), ((128, 500), (64, 250))], 60, 72.0),
        ("big", [4, 8], [((150, 2000), (75, 1000))], 40, 75.0),
        ("3level", [4, 4, 8], [((64, 64), (100, 500), (64, 250))], 60, 72.0),   # (a colour template below 64 features would put the level on the reference's 8-bit path, which asserts <= 63 for the normals too: LL.cpp:1457)
    ]
    for name, T, specs, n, thr in cases:
        od = lo.OracleDetector(64, T)
        pyr = od.quantize_pyramid(rgb, dep)
        banks = [synth.make_planted_bank(40 + k, n, [(p[0], p[1]) for p in pyr], T, nf) for k, nf in enumerate(specs)]
        raws, cands, evals = [], 0, 0
        for ci, b in enumerate(banks):
            raw, st = oracle_matches(od, rgb, dep, b, T, thr, ci)
            raws.append(raw); cands += st["coarse_candidates"]; evals += st["local_evals"]
        want = lo.canonical_sort_unique(np.concatenate(raws))
        assert len(want) > 0 and cands > n, name
        for paths in PATHS:
            det = detector_on(lm, paths, 64, T, device=0)
            for ci, b in enumerate(banks):
                det.addClassPacked("c%d" % ci, *b)
            same_records(det.matchArray([rgb, dep], thr, ["c%d" % ci for ci in range(len(banks))]), want)
            expect_paths(det, paths, tiles_possible=len(T) == 2, levels=len(T))
            tm = det.lastTimings()
            assert (tm["coarse_candidates"], tm["local_evals"]) == (cands, evals), (name, paths)
            # streamed in a batch of four: the same lists
            for _ in range(4):
                det.submitFrame([rgb, dep], thr, ["c%d" % ci for ci in range(len(banks))])
            for _ in range(4):
                same_records(det.collect(), want)


def test_reference_feature_ceiling_8191_per_modality(lm):
    """The reference's ceiling (LL.cpp:1291, 1816: 8191 features per modality = 32764 < 2^15 per u16 sum): template entries of 8191 + 8191 features at
    BOTH pyramid levels — 16382 per entry, 2 features short of what the 14-bit counters of k_local_bits<10, .> / k_coarse_bits<10, .> hold — through
    every kernel path and the oracle.  The features are those of planted 150- / 75-feature templates repeated, so that the scores keep their
    structure (candidates around the planted position, raw sums up to 4 x 16382 = 65528)."""
    W, H, T = 640, 480, [4, 8]
    rgb, dep = synth.make_frame(31, W, H, 40)
    od = lo.OracleDetector(64, T)
    pyr = od.quantize_pyramid(rgb, dep)
    feat, offs, wh = synth.make_planted_bank(77, 10, [(p[0], p[1]) for p in pyr], T, (150, 75))
    parts, noffs = [], [0]
    for k in range(len(offs) - 1):
        f = feat[offs[k]:offs[k + 1]]
        f = np.tile(f, ((8191 + len(f) - 1) // len(f), 1))[:8191]
        parts.append(f); noffs.append(noffs[-1] + len(f))
    big = (np.ascontiguousarray(np.concatenate(parts), dtype=np.int32), np.asarray(noffs, np.int32), wh)
    thr = 70.0
    raw, st = oracle_matches(od, rgb, dep, big, T, thr, 0)
    want = lo.canonical_sort_unique(raw)
    assert len(want) > 0 and st["coarse_candidates"] >= 10
    for paths in PATHS:
        det = detector_on(lm, paths, 64, T, device=0)
        det.addClassPacked("big", *big)
        same_records(det.matchArray([rgb, dep], thr, ["big"]), want)
        expect_paths(det, paths)
        tm = det.lastTimings()
        assert (tm["coarse_candidates"], tm["local_evals"], tm["matches_pre_unique"]) == (st["coarse_candidates"], st["local_evals"], len(raw)), paths
        for _ in range(3):
            det.submitFrame([rgb, dep], thr, ["big"])
        for _ in range(3):
            same_records(det.collect(), want)


def _random_geometries(seed, count):
    """Seeded random frame sizes / pyramids / feature counts / thresholds the reference accepts (rows and columns of every level multiples of its T,
    LL.cpp:1217-1218; rows x columns a multiple of 16, LL.cpp:1136): widths that are not multiples of 16 (no pixel tiles), odd steps, one to three levels."""
    import math
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        L = int(rng.choice([1, 2, 2, 2, 3]))
        T = [int(rng.choice([2, 4, 5, 8])) for _ in range(L)]
        T[-1] = int(rng.choice([4, 5, 8, 16]))
        base = 1
        for l, t in enumerate(T):
            base = math.lcm(base, t << l)
        base = math.lcm(base, 4 << (L - 1))
        a = int(rng.integers(max(1, 300 // base), max(2, 760 // base) + 1)); b = int(rng.integers(max(1, 220 // base), max(2, 560 // base) + 1))
        W, H = base * a, base * b
        if W < 300 or H < 220 or W > 800 or H > 600 or any(((W >> l) * (H >> l)) % 16 for l in range(L)):
            continue
        nf0 = int(rng.choice([24, 48, 64, 96, 150]))
        out.append((W, H, T, tuple(max(8, nf0 >> l) for l in range(L)), float(rng.choice([55.0, 65.0, 75.0])), int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("case", _random_geometries(2025, 14), ids=lambda c: "%dx%d-T%s-nf%d" % (c[0], c[1], "_".join(map(str, c[2])), c[3][0]))
def test_random_geometries_equal_the_oracle(lm, case):
    """Differential test over geometries nobody chose by hand: the default path and two others drawn per case against the oracle, records and counts."""
    W, H, T, nfeat, thr, sd = case
    rgb, dep = synth.make_frame(sd % 1000, W, H, 30)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    bank = synth.make_planted_bank(sd % 997, 40, [(p[0], p[1]) for p in pyr], T, nfeat)
    raw, st = oracle_matches(od, rgb, dep, bank, T, thr, 0)
    want = lo.canonical_sort_unique(raw)
    rng = np.random.default_rng(sd)
    for paths in [PATHS[0]] + [PATHS[i] for i in rng.choice(np.arange(1, len(PATHS)), 2, replace=False)]:
        det = detector_on(lm, paths, nfeat[0], T, device=0)
        det.addClassPacked("o", *bank)
        same_records(det.matchArray([rgb, dep], thr, ["o"]), want)
        tm = det.lastTimings()
        assert (tm["coarse_candidates"], tm["local_evals"], tm["matches_pre_unique"]) == (st["coarse_candidates"], st["local_evals"], len(raw)), (case, paths)
        for l in range(len(T)):                               # the quantised maps the match ran on
            assert np.array_equal(det.readStage(l, 0).reshape(H >> l, W >> l), pyr[l][0]) and np.array_equal(det.readStage(l, 1).reshape(H >> l, W >> l), pyr[l][1]), (case, l)
        for _ in range(3):                                    # and through the stream, three frames sharing a launch
            det.submitFrame([rgb, dep], thr, ["o"])
        for _ in range(3):
            same_records(det.collect(), want)


def test_sharded_equals_unsharded_on_one_device(lm):
    """N logical shards on one device through the same slice + merge code the multi-GPU path uses."""
    W, H, T, nfeat = 640, 480, [4, 8], (150, 75)
    rgb, dep = synth.make_frame(13, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    a = synth.make_planted_bank(31, 130, [(p[0], p[1]) for p in pyr], T, nfeat)
    b = synth.make_planted_bank(32, 77, [(p[0], p[1]) for p in pyr], T, nfeat)
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("a", *a)
    det.addClassPacked("b", *b)
    whole = det.matchArray([rgb, dep], 75.0, ["b", "a"])
    assert len(whole) > 0
    for world in (2, 3, 8):
        parts = []
        det.setFrame([rgb, dep])
        for distinct in (False, True):                          # raw pre-unique records / without exact duplicates (what the gather ships)
            parts = []
            for r in range(world):
                det.setShard(r, world)
                parts.append(det.matchResident(75.0, ["b", "a"], sort_unique=False, distinct=distinct))
            det.setShard(0, 1)
            merged = lm.merge_matches(np.concatenate(parts))
            assert merged.tobytes() == whole.tobytes()
        raw = det.matchResident(75.0, ["b", "a"], sort_unique=False)
        dis = det.matchResident(75.0, ["b", "a"], sort_unique=False, distinct=True)
        key = lambda r: set(zip(r["x"].tolist(), r["y"].tolist(), r["similarity"].tolist(), r["class_index"].tolist(), r["template_id"].tolist()))
        assert key(raw) == key(dis) and len(dis) == len(key(dis)) <= len(raw)


def test_device_exchange_merges_shards_like_the_host(lm):
    """SURVEY §8e on the device: W detectors stand in for W ranks of one job (same frame, shard r of W each); their
    packed blocks are concatenated the way an all-gather lays them out and every "rank" merges them with the ranking
    kernel.  The result must be the unsharded Detector.match list — order, adjacent-unique and all — on every rank."""
    import torch
    W_, H_, T, nfeat = 640, 480, [4, 8], (150, 75)
    rgb, dep = synth.make_frame(13, W_, H_)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    banks = [synth.make_planted_bank(31 + i, n, [(p[0], p[1]) for p in pyr], T, nfeat) for i, n in enumerate((130, 77))]
    ids = ["b", "a"]

    def make():
        d = lm.Detector(nfeat[0], T, device=0)
        for c, b in zip("ab", banks):
            d.addClassPacked(c, *b)
        d.setFrame([rgb, dep])
        return d
    ref = make()
    lib = lm.load_library()
    for thr, cap in ((75.0, 4096), (75.0, 8192), (60.0, 8192), (99.5, 256)):
        whole = ref.matchResident(thr, ids)
        assert len(whole) > 0 or thr > 99
        for world in (1, 2, 3, 8):
            dets = [make() for _ in range(world)]
            nb = lib.lm_exchange_block_bytes(cap)
            send = [torch.zeros(nb, dtype=torch.uint8, device="cuda:0") for _ in range(world)]
            for r, d in enumerate(dets):
                d.setShard(r, world)
                d.submit(thr, ids)
                d.exchangePack(send[r].data_ptr(), cap)
            streams = [torch.cuda.ExternalStream(d.exchangeStream(), device="cuda:0") for d in dets]
            for st in streams:
                st.synchronize()
            recv = torch.cat(send)
            counts = [int(b[:4].cpu().numpy().view(np.uint32)[0]) for b in send]
            pre = ref.matchResident(thr, ids, sort_unique=False, distinct=True)
            assert sum(counts) == len(pre), (counts, len(pre))
            into = np.empty(world * cap, lm.MATCH_DTYPE)
            for r, d in enumerate(dets):
                d.exchangeMerge(recv.data_ptr(), world, cap)
                got, failed = d.exchangeCollect() if r % 2 == 0 else d.exchangeCollectInto(into)
                assert failed == 0 and got.tobytes() == whole.tobytes(), (thr, cap, world, len(got), len(whole))
    # a rank with more distinct records than the former 8192-record limit of a block: blocks of up to
    # lm_exchange_max_capacity() records, several LDS tiles per run in the ranking kernel
    assert lib.lm_exchange_max_capacity() >= 65536
    big_thr = None
    for thr in (55.0, 50.0, 45.0, 40.0, 35.0):
        if len(ref.matchResident(thr, ids, sort_unique=False, distinct=True)) > 10000:
            big_thr = thr
            break
    assert big_thr is not None, "no threshold yields more than 10000 distinct records"
    whole = ref.matchResident(big_thr, ids)
    n_distinct = len(ref.matchResident(big_thr, ids, sort_unique=False, distinct=True))
    fit = 1 << int(np.ceil(np.log2(n_distinct)))
    assert 8192 < fit <= 65536, n_distinct
    for world, cap in ((1, fit), (2, min(65536, 2 * fit)), (3, 65536)):
        dets = [make() for _ in range(world)]
        nb = lib.lm_exchange_block_bytes(cap)
        send = [torch.zeros(nb, dtype=torch.uint8, device="cuda:0") for _ in range(world)]
        for r, d in enumerate(dets):
            d.setShard(r, world)
            d.matchResident(big_thr, ids)          # a threshold this low outgrows the default candidate buffer: the synchronous call grows it
            d.submit(big_thr, ids); d.exchangePack(send[r].data_ptr(), cap)
            torch.cuda.ExternalStream(d.exchangeStream(), device="cuda:0").synchronize()
        counts = [int(b[:4].cpu().numpy().view(np.uint32)[0]) for b in send]
        assert world > 1 or counts[0] > 8192, counts
        recv = torch.cat(send)
        for d in dets:
            d.exchangeMerge(recv.data_ptr(), world, cap)
            got, failed = d.exchangeCollect()
            assert failed == 0, (failed, counts, big_thr, cap, world)
            assert got.tobytes() == whole.tobytes(), (big_thr, cap, world, len(got), len(whole))
    # many ranks (more runs than one boundary group of the ranking kernel holds, most of them empty or tiny): one detector
    # produces the blocks shard by shard, then merges them
    d = make()
    whole = ref.matchResident(75.0, ids)
    for world, cap in ((70, 8192), (150, 4096), (300, 256)):          # 300 ranks for 207 templates: shards without any work
        nb = lib.lm_exchange_block_bytes(cap)
        send = [torch.zeros(nb, dtype=torch.uint8, device="cuda:0") for _ in range(world)]
        for r in range(world):
            d.setShard(r, world); d.submit(75.0, ids); d.exchangePack(send[r].data_ptr(), cap)
            d.collect(sort_unique=False)                                   # retires the frame without the exchange
        recv = torch.cat(send)
        d.setShard(0, world); d.submit(75.0, ids); d.exchangePack(send[0].data_ptr(), cap)
        d.exchangeMerge(recv.data_ptr(), world, cap)
        got, failed = d.exchangeCollect()
        assert failed == 0 and got.tobytes() == whole.tobytes(), (world, cap, len(got), len(whole))
    d.setShard(0, 1)
    # blocks too small for the records: every rank reports the same need, nothing is returned
    dets = [make() for _ in range(2)]
    nb = lib.lm_exchange_block_bytes(256)
    send = [torch.zeros(nb, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    for r, d in enumerate(dets):
        d.setShard(r, 2); d.submit(60.0, ids); d.exchangePack(send[r].data_ptr(), 256)
        torch.cuda.ExternalStream(d.exchangeStream(), device="cuda:0").synchronize()
    recv = torch.cat(send)
    needs = []
    for d in dets:
        d.exchangeMerge(recv.data_ptr(), 2, 256)
        got, failed = d.exchangeCollect()
        assert got is None and failed > 256
        needs.append(failed)
    assert needs[0] == needs[1]
    # protocol errors are loud
    d = make()
    with pytest.raises(RuntimeError, match="no frame in flight"):
        d.exchangePack(send[0].data_ptr(), 256)
    d.submit(75.0, ids)
    with pytest.raises(RuntimeError, match="power of two"):
        d.exchangePack(send[0].data_ptr(), 300)
    with pytest.raises(RuntimeError, match="precede"):
        d.exchangeMerge(recv.data_ptr(), 2, 256)
    with pytest.raises(RuntimeError, match="not exchanged"):
        d.exchangeCollect()
    assert len(d.collect()) == len(ref.matchResident(75.0, ids))


def test_pipelined_submit_collect_equals_synchronous(lm):
    """Stream mode: the maximum number of frames in flight (front end of k+2, coarse pass / refinement of k+1 on their
    streams while the host collects k) returns exactly what the synchronous calls return, frame by frame, also when the
    stream wraps around the result slots; one more submit without a collect is refused."""
    W, H, T, nfeat = 640, 480, [4, 8], (150, 75)
    frames = [synth.make_frame(50 + i, W, H) for i in range(5)]
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(*frames[0])
    bank = synth.make_planted_bank(61, 150, [(p[0], p[1]) for p in pyr], T, nfeat)
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("o", *bank)
    want = []
    for i, f in enumerate(frames):
        det.storeFrame(i, f)
        want.append(det.matchArray(list(f), 70.0, ["o"]))
    assert len(want[0]) > 0
    depth = lm.load_library().lm_detector_max_in_flight()
    assert depth >= 3
    got, order = [], []
    for k in range(depth):
        det.selectFrame(k % 5); det.submit(70.0, ["o"]); order.append(k % 5)
    with pytest.raises(RuntimeError, match="in flight"):
        det.submit(70.0, ["o"])
    got.append(det.collect())
    for k in range(depth, depth + 9):                         # keep the pipeline full for a while
        det.selectFrame(k % 5); det.submit(70.0, ["o"]); order.append(k % 5)
        got.append(det.collect())
    for _ in range(depth - 1):
        got.append(det.collect())
    with pytest.raises(RuntimeError, match="no frame in flight"):
        det.collect()
    assert len(got) == len(order)
    for g, fi in zip(got, order):
        assert g.tobytes() == want[fi].tobytes()


@pytest.mark.parametrize("mode", ["no_helpers", "one_cpu", "many_helpers"])
def test_stream_with_and_without_helper_threads(lm, mode):
    """The helper threads of the streamed path (HostPool: sliced staging copy, result lists of a batch's later frames) are an optimisation the results
    never depend on: none at all (LM_HOST_THREADS=0), a process that may run on ONE CPU (the pool shrinks to nothing: never more helpers than the CPUs
    the process may use leave free), and more helpers asked for than the default — the same lists as the synchronous call, batches shared."""
    W, H, T, nfeat = 640, 480, [4, 8], (150, 75)
    frames = [synth.make_frame(170 + i, W, H) for i in range(10)]
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(*frames[0])
    bank = synth.make_planted_bank(72, 100, [(p[0], p[1]) for p in pyr], T, nfeat)
    ref = lm.Detector(nfeat[0], T, device=0)
    ref.addClassPacked("o", *bank)
    want = [ref.matchArray(list(f), 70.0, ["o"]) for f in frames]
    old_env, old_aff = os.environ.get("LM_HOST_THREADS"), os.sched_getaffinity(0)
    try:
        if mode == "no_helpers": os.environ["LM_HOST_THREADS"] = "0"
        if mode == "many_helpers": os.environ["LM_HOST_THREADS"] = "7"
        if mode == "one_cpu": os.sched_setaffinity(0, {min(old_aff)})
        det = lm.Detector(nfeat[0], T, device=0)                 # the pool is sized when the detector is created / first used
        det.addClassPacked("o", *bank)
        det.setBatch(4)
        for rep in range(2):
            for f in frames: det.submitFrame(list(f), 70.0, ["o"])
            for w in want:
                assert det.collect().tobytes() == w.tobytes()
        del det
    finally:
        os.sched_setaffinity(0, old_aff)
        if old_env is None: os.environ.pop("LM_HOST_THREADS", None)
        else: os.environ["LM_HOST_THREADS"] = old_env


def test_live_stream_ingest_equals_synchronous_and_the_oracle(lm):
    """SURVEY §8f N4 proper: a NEW host frame per step (linemod_ros/detect.py:83-138, linemod_and_levelup_test.py:314-327)
    through lm_detector_submit_frame — pinned ring + copy stream, up to lm_detector_max_in_flight() frames in flight, every frame different —
    returns, frame by frame, exactly what the oracle and the synchronous Detector.match return; also through the zero-copy
    ring buffers, the matchStream generator, mixed with parked-frame submits, and across a frame-size change."""
    W, H, T, nfeat = 640, 480, [4, 8], (150, 75)
    n_frames = 11
    frames = [synth.make_frame(150 + i, W, H) for i in range(n_frames)]
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(*frames[0])
    bank = synth.make_planted_bank(71, 120, [(p[0], p[1]) for p in pyr], T, nfeat)
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("o", *bank)
    want = [det.matchArray(list(f), 70.0, ["o"]) for f in frames]
    for i in (0, 5, 10):                                        # the synchronous call itself against the oracle
        raw, _ = oracle_matches(od, frames[i][0], frames[i][1], bank, T, 70.0)
        same_records(want[i], lo.canonical_sort_unique(raw))
    assert len(want[0]) > 0 and len({w.tobytes() for w in want}) > 1, "frames must differ"
    depth = lm.load_library().lm_detector_max_in_flight()
    # (1) host arrays, the ring kept full, wrapping around it several times
    got = []
    for k, f in enumerate(frames):
        scratch = (f[0].copy(), f[1].copy())
        det.submitFrame(scratch, 70.0, ["o"])
        scratch[0][:] = 0; scratch[1][:] = 0                     # borrowed only during the call
        if k >= depth - 1:
            got.append(det.collect())
    with pytest.raises(RuntimeError, match="in flight"):
        det.setFrame(list(frames[0]))                            # the blocking upload would pull the rug from under frames in flight
    while len(got) < n_frames:
        got.append(det.collect())
    for g, w in zip(got, want):
        assert g.tobytes() == w.tobytes()
    assert det.lastTimings()["h2d_ms"] > 0
    # (2) zero-copy: the caller writes into the pinned ring entry the next submit uploads from
    got = []
    for k, f in enumerate(frames):
        rgb_buf, dep_buf = det.ingestBuffers(W, H)
        rgb_buf[:] = f[0]; dep_buf[:] = f[1]
        det.submitFrame((rgb_buf, dep_buf), 70.0, ["o"])
        if k >= 2:
            got.append(det.collect())
    while len(got) < n_frames:
        got.append(det.collect())
    for g, w in zip(got, want):
        assert g.tobytes() == w.tobytes()
    # (3) the generator form of the dataset loop, at every depth
    for dpt in (1, 2, depth):
        for g, w in zip(det.matchStream(iter(frames), 70.0, ["o"], depth=dpt), want):
            assert g.tobytes() == w.tobytes()
    # (4) streamed and parked frames interleaved (the front end's source pointers alternate)
    det.storeFrame(0, frames[3])
    det.submitFrame(frames[1], 70.0, ["o"])
    det.selectFrame(0); det.submit(70.0, ["o"])
    det.submitFrame(frames[2], 70.0, ["o"])
    for i in (1, 3, 2):
        assert det.collect().tobytes() == want[i].tobytes()
    # (5) frame-size change: refused with frames in flight, fine once they are collected
    small = synth.make_frame(7, 320, 240, 14)
    det.submitFrame(frames[4], 70.0, ["o"])
    with pytest.raises(RuntimeError, match="in flight"):
        det.submitFrame(small, 70.0, ["o"])
    with pytest.raises(RuntimeError, match="in flight"):
        det.storeFrame(1, small); det.selectFrame(1)
    assert det.collect().tobytes() == want[4].tobytes()
    det.submitFrame(small, 70.0, ["o"])
    s1 = det.collect()
    assert s1.tobytes() == det.matchArray(list(small), 70.0, ["o"]).tobytes()
    assert det.matchArray(list(frames[6]), 70.0, ["o"]).tobytes() == want[6].tobytes()
    with pytest.raises(RuntimeError):
        det.submitFrame((frames[0][0], frames[0][1][:100]), 70.0, ["o"])
    # (6) frames per launch: the streamed frames of a batch share ONE front end / coarse / refinement / duplicate-removal launch
    # (lm_detector_set_batch); per frame the records, the candidate count and the evaluation count are those of the synchronous
    # call whatever the batch size, also when a batch is cut short by flush(), by collect(), by the end of the slot ring, by a
    # change of threshold or of the class list
    stats = []
    for f in frames:
        det.matchArray(list(f), 70.0, ["o"])
        tm = det.lastTimings()
        assert tm["batch_frames"] == 1
        stats.append((tm["coarse_candidates"], tm["local_evals"], tm["local_bytes"], tm["matches_pre_unique"]))
    want65 = [det.matchArray(list(f), 65.0, ["o"]) for f in frames[:3]]
    det.addClassPacked("p", *synth.make_planted_bank(72, 40, [(p[0], p[1]) for p in pyr], T, nfeat))
    want_po = [det.matchArray(list(f), 70.0, ["p", "o"]) for f in frames[:3]]
    assert det.getBatch() == 8
    for nbatch in (1, 2, 3, 4, 5, 8, 4):
        det.setBatch(nbatch)
        det.setBatchQueue(0 if nbatch != 4 else 2)                 # 0: full batches only (deterministic sizes); 2 = default: early launches while the GPU's queue is short
        seen = set()
        for k, f in enumerate(frames):
            det.submitFrame(f, 70.0, ["o"])
            if k == 8:
                det.flush()
        for k in range(n_frames):
            assert det.collect().tobytes() == want[k].tobytes(), (nbatch, k)
            tm = det.lastTimings()
            assert (tm["coarse_candidates"], tm["local_evals"], tm["local_bytes"], tm["matches_pre_unique"]) == stats[k], (nbatch, k)
            assert 1 <= tm["batch_frames"] <= nbatch
            seen.add(tm["batch_frames"])
        assert nbatch in seen or nbatch == 4, (nbatch, seen)
        # threshold and class list change inside what would be one batch
        det.submitFrame(frames[0], 70.0, ["o"]); det.submitFrame(frames[1], 65.0, ["o"]); det.submitFrame(frames[2], 65.0, ["o"])
        det.submitFrame(frames[0], 70.0, ["p", "o"]); det.submitFrame(frames[1], 70.0, ["p", "o"]); det.submitFrame(frames[2], 70.0, ["o"])
        exp = [want[0], want65[1], want65[2], want_po[0], want_po[1], want[2]]
        for k, e in enumerate(exp):
            assert det.collect().tobytes() == e.tobytes(), (nbatch, "mixed", k)
    with pytest.raises(RuntimeError):
        det.setBatch(9)
    det.setBatch(8)


def test_config1_size_2k_templates_bit_exact(lm):
    """BASELINE configs[1]: 1 object x 2k templates, 640x480 — compared directly (the C oracle takes
    ~0.3 s) plus size-independent properties: threshold monotonicity and bank-permutation invariance."""
    W, H, T, nfeat, n = 640, 480, [4, 8], (150, 75), 2000
    rgb, dep = synth.make_frame(0, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    bank = synth.make_planted_bank(1234, n, [(p[0], p[1]) for p in pyr], T, nfeat)
    want, st = oracle_matches(od, rgb, dep, bank, T, 75.0)
    for paths in PATHS[::-1]:                                     # every kernel path at the bench's size; the default (bit planes) last and kept
        det = detector_on(lm, paths, nfeat[0], T, device=0)
        det.addClassPacked("obj", *bank)
        got = det.matchArray([rgb, dep], 75.0, ["obj"])
        same_records(got, lo.canonical_sort_unique(want))
        expect_paths(det, paths)
        tm = det.lastTimings()
        assert tm["coarse_candidates"] == st["coarse_candidates"] and tm["local_evals"] == st["local_evals"], paths
    hi = det.matchArray([rgb, dep], 85.0, ["obj"])
    key = lambda r: set(zip(r["x"].tolist(), r["y"].tolist(), r["similarity"].tolist(), r["template_id"].tolist()))
    assert key(hi) <= key(got) and all(hi["similarity"] >= 85.0)
    # permuting the template pyramids permutes template ids and nothing else
    feat, offs, wh = bank
    perm = np.random.default_rng(3).permutation(n)
    E = 4
    pf, po, pw = [], [0], []
    for p in perm:
        for e in range(E):
            k = p * E + e
            pf.append(feat[offs[k]:offs[k + 1]]); po.append(po[-1] + offs[k + 1] - offs[k]); pw.append(wh[k])
    det2 = lm.Detector(nfeat[0], T, device=0)
    det2.addClassPacked("obj", np.concatenate(pf), np.asarray(po, np.int32), np.asarray(pw, np.int32))
    det2.setFrame([rgb, dep])
    pre2 = det2.matchResident(75.0, ["obj"], sort_unique=False)
    back = pre2.copy()
    back["template_id"] = perm[pre2["template_id"]]
    assert as_multiset(back, ["x", "y", "similarity", "template_id"]) == as_multiset(want, ["x", "y", "sim", "tid"])


def test_config4_shard_size(lm):
    """One GPU's share of BASELINE configs[4]: 1280x960, 11250 template pyramids — compared record by record with the
    oracle (its C loops on 32 host threads take a fraction of a second)."""
    W, H, T, nfeat, n = 1280, 960, [4, 8], (150, 75), 11250
    rgb, dep = synth.make_frame(0, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    bank = synth.make_planted_bank(99, n, [(p[0], p[1]) for p in pyr], T, nfeat)
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("obj", *bank)
    feat, offs, wh = bank
    lms, sizes = od.linear_memories(rgb, dep)
    raw, st = lo.match_bank_c(lo.PackedBank(n, 2, feat, offs, wh), lms, sizes, T, 75.0, nthreads=32)
    raw["cls"] = 0
    got = det.matchArray([rgb, dep], 75.0, ["obj"])
    same_records(got, lo.canonical_sort_unique(raw))
    tm = det.lastTimings()
    assert tm["coarse_candidates"] == st["coarse_candidates"] > 50000 and tm["local_evals"] == st["local_evals"]


def test_config3_size_eight_shards_through_the_device_exchange(lm):
    """BASELINE configs[3]: 8 objects x 2000 templates at 640x480, one object's worth of templates per GPU.  The oracle pins the
    whole 16k-template result (C loops, 32 host threads); eight shards of it — searched one after the other on this GPU, packed,
    laid out like an all-gather and merged by the ranking kernel — must be that list, bit for bit."""
    import torch
    W, H, T, nfeat, per, world, cap = 640, 480, [4, 8], (150, 75), 2000, 8, 4096
    rgb, dep = synth.make_frame(0, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    ids = ["obj%02d" % o for o in range(world)]
    banks = [synth.make_planted_bank(1234 + o, per, [(p[0], p[1]) for p in pyr], T, nfeat) for o in range(world)]
    det = lm.Detector(nfeat[0], T, device=0)
    lms, sizes = od.linear_memories(rgb, dep)
    raws = []
    for o, (cid, bank) in enumerate(zip(ids, banks)):
        det.addClassPacked(cid, *bank)
        feat, offs, wh = bank
        raw, _ = lo.match_bank_c(lo.PackedBank(per, 2, feat, offs, wh), lms, sizes, T, 75.0, nthreads=32)
        raw["cls"] = o
        raws.append(raw)
    want = lo.canonical_sort_unique(np.concatenate(raws))
    det.setFrame([rgb, dep])
    whole = det.matchResident(75.0, ids)
    same_records(whole, want)
    assert len(whole) > 10000
    nb = lm.load_library().lm_exchange_block_bytes(cap)
    send = [torch.zeros(nb, dtype=torch.uint8, device="cuda:0") for _ in range(world)]
    for r in range(world):
        det.setShard(r, world); det.submit(75.0, ids); det.exchangePack(send[r].data_ptr(), cap)
        part = det.collect(sort_unique=False, distinct=True)
        assert set(part["class_index"].tolist()) == {r}                    # shard r of 8 = object r
    recv = torch.cat(send)
    det.setShard(3, world); det.submit(75.0, ids); det.exchangePack(send[3].data_ptr(), cap)
    det.exchangeMerge(recv.data_ptr(), world, cap)
    got, failed = det.exchangeCollect()
    det.setShard(0, 1)
    assert failed == 0 and got.tobytes() == whole.tobytes()


# ---------------------------------------------------------------------------------------------
# addTemplate / YAML through the product
# ---------------------------------------------------------------------------------------------
def test_add_template_reproduces_reference_golden(lm, tmp_path):
    rgb, dep, mask = load_bgr("train_rgb.png"), load_u16("train_dep.png"), load_gray("train_mask.png")
    det = lm.Detector(device=0)                      # Detector(): 63 features, T={5,8}
    assert det.addTemplate([rgb, dep], "06_template", mask) == 0
    _, _, _, pyr = lo.read_class_yaml(os.path.join(GOLDEN, "writeClasses_06_template.yaml"))
    for a, b in zip(det.getTemplates("06_template", 0), pyr[0]):
        assert (a.width, a.height, a.pyramid_level) == (b.width, b.height, b.pyramid_level)
        assert np.array_equal(a.features, b.features)
    # a frame without enough features fails with -1 and adds nothing (LL.cpp:1964-1966)
    assert det.addTemplate([np.zeros_like(rgb), np.zeros_like(dep)], "06_template", mask) == -1
    assert det.numTemplates("06_template") == 1
    # no-mask variant equals the oracle
    od = lo.OracleDetector()
    srgb, sdep = synth.make_frame(9, 320, 240, 10)
    t_o = od.addTemplate([srgb, sdep], "s", None)
    t_g = det.addTemplate([srgb, sdep], "s", np.zeros((0, 0), np.uint8))
    assert t_o == t_g == 0
    for a, b in zip(det.getTemplates("s", 0), od.class_templates["s"][0]):
        assert (a.width, a.height) == (b.width, b.height) and np.array_equal(a.features, b.features)
    # with an object mask the selection runs on the device (train.hip) — the golden above went through it; it, the host selection
    # (LM_TRAIN_HOST=1) and a grey mask (minima / differences of cv::erode / cv::subtract: always the host) all equal the oracle
    yy, xx = np.mgrid[0:240, 0:320]
    ell = (((xx - 160) / 120.0) ** 2 + ((yy - 120) / 90.0) ** 2 <= 1.0).astype(np.uint8) * 255
    ell[100:140, 150:170] = 0
    grey = ell.copy()
    grey[ell > 0] = 200
    grey[60:180:7, 80:240:5] = 90
    for name, m, envs in (("ell", ell, ("0", "1")), ("grey", grey, ("0",))):
        assert od.addTemplate([srgb, sdep], name, m) == 0
        want = od.class_templates[name][0]
        for env in envs:
            os.environ["LM_TRAIN_HOST"] = env
            try:
                dd = lm.Detector(device=0)
                assert dd.addTemplate([srgb, sdep], name, m) == 0
            finally:
                os.environ.pop("LM_TRAIN_HOST", None)
            for a, b in zip(dd.getTemplates(name, 0), want):
                assert (a.width, a.height) == (b.width, b.height) and np.array_equal(a.features, b.features), (name, env)
    # writeClasses -> readClasses round trip (oracle reader parses the product's YAML too)
    det.writeClasses(str(tmp_path / "%s.yaml"))
    _, mods, levels, pyr2 = lo.read_class_yaml(str(tmp_path / "06_template.yaml"))
    assert mods == ["ColorGradient", "DepthNormal"] and levels == 2
    assert np.array_equal(pyr2[0][0].features, pyr[0][0].features)
    det2 = lm.Detector(device=0)
    det2.readClasses(["06_template", "s"], str(tmp_path / "%s.yaml"))
    assert det2.classIds() == ["06_template", "s"]
    with pytest.raises(RuntimeError):
        det2.readClasses(["s"], str(tmp_path / "%s.yaml"))             # already present, LL.cpp:2059
    with pytest.raises(RuntimeError):
        lm.Detector([5, 8, 8], device=0).readClasses(["s"], str(tmp_path / "%s.yaml"))   # pyramid_levels, LL.cpp:2052


def test_bind_near_device_stays_within_the_allowed_cpus(lm):
    """lm_bind_thread_near_device: the GPU's local CPUs (sysfs), never more than the thread was allowed before."""
    before = os.sched_getaffinity(0)
    try:
        cpus = lm.bind_near_device(0)
        if cpus is None:
            pytest.skip("sysfs does not list the GPU's local CPUs on this box")
        now = os.sched_getaffinity(0)
        assert now and now <= before
        listed = set()
        for part in cpus.split(","):
            a, _, b = part.partition("-")
            listed.update(range(int(a), int(b or a) + 1))
        assert now <= listed
    finally:
        os.sched_setaffinity(0, before)


def test_detector_params_write_read_round_trip(lm, tmp_path):
    """Detector::write / read (LL.cpp:2013-2041): parameters survive the YAML, read() clears the classes, and a detector
    configured from the file matches like the one that wrote it."""
    rgb, dep, mask = load_bgr("train_rgb.png"), load_u16("train_dep.png"), load_gray("train_mask.png")
    a = lm.Detector(127, [4, 8], device=0)
    assert a.addTemplate([rgb, dep], "c", mask) == 0
    path = str(tmp_path / "detector.yaml")
    a.write(path)
    text = open(path).read()
    assert "pyramid_levels: 2" in text and "T: [ 4, 8 ]" in text and "type: ColorGradient" in text and "weak_threshold: 10." in text
    assert "num_features: 127" in text and "type: DepthNormal" in text and "distance_threshold: 2000" in text
    b = lm.Detector(device=0)                                    # Detector(): 63 features, T = {5, 8}
    assert b.addTemplate([rgb, dep], "c", mask) == 0
    b.read(path)
    assert b.numClasses() == 0 and b.getT(0) == 4 and b.pyramidLevels() == 2
    assert b.addTemplate([rgb, dep], "c", mask) == 0
    for x, y in zip(a.getTemplates("c", 0), b.getTemplates("c", 0)):
        assert (x.width, x.height) == (y.width, y.height) and np.array_equal(x.features, y.features)
    assert len(x.features) > 0
    with pytest.raises(RuntimeError):
        b.read(str(tmp_path / "missing.yaml"))


def test_packed_bank_file_round_trip(lm, tmp_path):
    """the packed binary bank holds what the per-class YAML files hold (LL.cpp:2043-2146) — every template
    of every class comes back identical, a detector loaded from it matches bit-exactly, and a file written by the test
    itself (layout of csrc/bank_file.cpp) is read the same way."""
    from test_host_cpu import _bank_file_bytes
    rgb, dep = load_bgr("0000_rgb.png"), load_u16("0000_dep.png")
    a = lm.Detector(127, [5, 8], device=0)
    a.readClasses(["06_template"], os.path.join(GOLDEN, "bank127_%s.yaml.gz"))
    rng = np.random.default_rng(11)
    P, E = 7, 4                                         # a second class: random pyramids, one empty template, negative coordinates
    counts = rng.integers(0, 130, P * E); counts[5] = 0
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    feats = np.stack([rng.integers(-3, 200, offs[-1]), rng.integers(-3, 150, offs[-1]), rng.integers(0, 8, offs[-1])], 1).astype(np.int32)
    wh = np.repeat(rng.integers(30, 200, (P * 2, 2)), 2, axis=0).astype(np.int32)   # both modalities of a level share the size
    a.addClassPacked("zz random", feats, offs, wh)
    path = tmp_path / "all.lmb"
    a.writeBank(path)
    info = lm.bank_file_info(path)
    assert info["class_ids"] == ["06_template", "zz random"] and info["num_pyramids"] == 89 + len(wh) // 4
    yaml_bytes = 0
    a.writeClasses(str(tmp_path / "%s.yaml"))
    for c in a.classIds():
        yaml_bytes += os.path.getsize(tmp_path / (c + ".yaml"))
    assert os.path.getsize(path) * 6 < yaml_bytes                       # 4 B per feature against ~45 text bytes

    b = lm.Detector(127, [5, 8], device=0)
    b.readBank(path)
    assert b.classIds() == a.classIds()
    for cid in a.classIds():
        assert a.numTemplates(cid) == b.numTemplates(cid)
        for tid in range(a.numTemplates(cid)):
            for x, y in zip(a.getTemplates(cid, tid), b.getTemplates(cid, tid)):
                assert (x.width, x.height, x.pyramid_level) == (y.width, y.height, y.pyramid_level)
                assert np.array_equal(x.features, y.features)
    for ids in ([], ["06_template"]):
        assert np.array_equal(b.matchArray([rgb, dep], 75.0, ids), a.matchArray([rgb, dep], 75.0, ids))
    assert len(a.matchArray([rgb, dep], 75.0, ["06_template"])) > 0

    # class filter (a rank that serves one object), double load, level mismatch, selective write
    c = lm.Detector(127, [5, 8], device=0)
    c.readBank(path, ["zz random"])
    assert c.classIds() == ["zz random"]
    with pytest.raises(RuntimeError, match="already present"):
        c.readBank(path)
    assert c.classIds() == ["zz random"]                                # nothing was added by the failed call
    with pytest.raises(RuntimeError, match="not in"):
        c.readBank(path, ["nope"])
    with pytest.raises(RuntimeError, match="pyramid levels"):
        lm.Detector([4, 4, 8], device=0).readBank(path)
    a.writeBank(tmp_path / "one.lmb", ["06_template"])
    assert lm.bank_file_info(tmp_path / "one.lmb")["class_ids"] == ["06_template"]
    with pytest.raises(RuntimeError):
        a.writeBank(tmp_path / "x.lmb", ["nope"])

    # a file produced outside the library
    P, E = 3, 4
    counts = rng.integers(1, 60, P * E)
    f2 = np.stack([rng.integers(0, 100, counts.sum()), rng.integers(0, 100, counts.sum()), rng.integers(0, 8, counts.sum())], 1)
    wh2 = np.repeat(rng.integers(30, 120, (P * 2, 2)), 2, axis=0)
    (tmp_path / "hand.lmb").write_bytes(_bank_file_bytes(2, [("hand", wh2, counts, f2)]))
    d = lm.Detector(127, [5, 8], device=0)
    d.readBank(tmp_path / "hand.lmb")
    k = 0
    for tid in range(P):
        for e, t in enumerate(d.getTemplates("hand", tid)):
            n = counts[tid * E + e]
            assert np.array_equal(t.features, f2[k:k + n]) and (t.width, t.height) == tuple(wh2[tid * E + e])
            k += n
    # y outside the 13-bit field: the packed format refuses, the YAML path does not
    e2 = lm.Detector(127, [5, 8], device=0)
    big = f2.astype(np.int32).copy(); big[0, 1] = 5000
    e2.addClassPacked("tall", big, np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), wh2.astype(np.int32))
    with pytest.raises(RuntimeError, match="packed format"):
        e2.writeBank(tmp_path / "tall.lmb")
    assert not os.path.exists(tmp_path / "tall.lmb")


# ---------------------------------------------------------------------------------------------
# poseRefine / ICP  (parity unpinned by the reference: GPU vs the oracle's Open3D restatement)
# ---------------------------------------------------------------------------------------------
K_CAM = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32).reshape(3, 3)


def _perturbed_scene(md, K, rot_deg, t_mm, seed):
    """Scene depth = model surface moved by a small SE(3) (re-rendered by forward splatting) + 1 mm noise."""
    rng = np.random.default_rng(seed)
    ys, xs = np.nonzero(md)
    z = md[ys, xs].astype(np.float64)
    P = np.stack([(xs - K[0, 2]) / K[0, 0] * z, (ys - K[1, 2]) / K[1, 1] * z, z], 1)
    a = np.radians(rot_deg)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    c = P.mean(0)
    Q = (P - c) @ Rz.T + c + np.asarray(t_mm, np.float64)
    u = np.rint(Q[:, 0] / Q[:, 2] * K[0, 0] + K[0, 2]).astype(int)
    v = np.rint(Q[:, 1] / Q[:, 2] * K[1, 1] + K[1, 2]).astype(int)
    sd = np.zeros(md.shape, np.float64)
    ok = (u >= 0) & (u < md.shape[1]) & (v >= 0) & (v < md.shape[0])
    sd[v[ok], u[ok]] = Q[ok, 2] + rng.normal(0, 1.0, int(ok.sum()))
    return np.clip(np.rint(sd), 0, 65535).astype(np.uint16)


def _check_pose(lm, sd, md, dx, dy, intended, tol=1e-4):
    R, t = np.eye(3, dtype=np.float32), np.array([0, 0, 1000], np.float32)
    pr = lm.poseRefine(device=0, scene_from_scene=intended)
    pr.process(sd, md, K_CAM, K_CAM, R, t, dx, dy)
    ref = lo.pose_refine(sd, md, K_CAM, K_CAM, R, t, dx, dy, scene_from_scene=intended)
    assert pr.info["n_source"] == ref["n_source"] and pr.info["n_target"] == ref["n_target"]
    assert pr.info["iterations"] == ref["iterations"]
    assert abs(pr.getResidual() - ref["residual"]) < 1e-6
    assert np.abs(pr.getR() - ref["R"]).max() < tol                       # rotation entries
    assert np.abs(pr.getT().ravel() - ref["t"]).max() / 1000.0 < tol      # translation in metres
    assert pr.getR().dtype == np.float64 and pr.getT().shape == (3, 1)
    return pr, ref


@pytest.mark.parametrize("seed,rot,tmm", [(1, 1.0, (2.0, -1.5, 3.0)), (2, -2.5, (-3.0, 2.0, -4.0)), (3, 0.0, (0.0, 0.0, 5.0))])
def test_pose_refine_intended_mode(lm, seed, rot, tmm):
    md = synth.synth_model_depth(seed)
    sd = _perturbed_scene(md, K_CAM.astype(np.float64), rot, tmm, seed)
    ys, xs = np.nonzero(md)
    pr, ref = _check_pose(lm, sd, md, int(xs.min()), int(ys.min()), True)
    assert ref["residual"] > 0.9 and ref["iterations"] >= 1               # a well-posed registration


def test_pose_refine_reference_fixture_images(lm):
    md, sd = load_u16("pose_depth_ren.png"), load_u16("pose_0003.png")
    ys, xs = np.nonzero(md)
    # verbatim (LL.cpp:109: target = model cloud) in its two well-posed regimes:
    scene_small = np.where(md > 0, md + 2, 0).astype(np.uint16)           # centroid offset 2 mm -> identity, fitness 1
    pr, ref = _check_pose(lm, scene_small, md, int(xs.min()) - 4, int(ys.min()) - 4, False)
    assert pr.getResidual() == 1.0 and ref["n_source"] == 1316
    scene_far = np.where(md > 0, md + 250, 0).astype(np.uint16)           # 250 mm: no correspondences -> init_guess, fitness 0
    pr, ref = _check_pose(lm, scene_far, md, int(xs.min()) - 4, int(ys.min()) - 4, False)
    assert pr.getResidual() == 0.0 and pr.info["iterations"] == 1
    # the real scene depth image of the fixture in intended mode (well-conditioned registration)
    _check_pose(lm, scene_small, md, int(xs.min()) - 4, int(ys.min()) - 4, True)


def test_pose_refine_rejects_window_outside_frame(lm):
    md = synth.synth_model_depth(4)
    pr = lm.poseRefine(device=0)
    pr.process(md, md, K_CAM, K_CAM, np.eye(3, dtype=np.float32), np.array([0, 0, 1000], np.float32), 620, 10)
    assert pr.getResidual() == -1 and pr.getR() is None                    # LL.cpp:52-55
    assert lo.pose_refine(md, md, K_CAM, K_CAM, np.eye(3), np.zeros(3), 620, 10)["residual"] == -1.0


def test_pose_refine_batch_equals_single_calls(lm):
    import linemodLevelup_pybind as mod
    mds, sds, xy = [], [], []
    base = synth.synth_model_depth(10)
    scene = _perturbed_scene(base, K_CAM.astype(np.float64), 1.5, (2.0, 1.0, -3.0), 10)
    for s in range(4):
        md = synth.synth_model_depth(10 + s)
        ys, xs = np.nonzero(md)
        mds.append(md); xy.append((int(xs.min()), int(ys.min())))
    xy.append((630, 470)); mds.append(mds[0])                               # one rejected hypothesis
    n = len(mds)
    Ks = np.tile(K_CAM.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1))
    ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
    res, ms = mod.pose_refine_batch(scene, K_CAM, mds, Ks, Rs, ts, xy, device=0, scene_from_scene=True)
    assert ms > 0 and res[-1]["residual"] == -1.0
    for i in range(n - 1):
        pr = mod.poseRefine(device=0, scene_from_scene=True)
        pr.process(scene, mds[i], K_CAM, K_CAM, Rs[i].reshape(3, 3), ts[i], xy[i][0], xy[i][1])
        # the members of a hypothesis' team (k_icp_team) are dealt out by the sizes of the clouds in the batch, so the 29 sums of an evaluation are
        # grouped differently in a batch of five and alone: equal to rounding, not bit for bit (the same batch is: see the slots test below)
        assert pr.info["iterations"] == res[i]["iterations"]
        assert np.abs(pr.getR() - res[i]["R"]).max() < 1e-11 and np.abs(pr.getT().ravel() - res[i]["t"]).max() < 1e-8


def test_icp_device_intermediates_match_oracle(lm):
    """Every device stage of poseRefine against the oracle's intermediates: init_guess, both voxel-down-sampled
    clouds (same order), the kNN normals (up to sign), in both target modes."""
    import linemodLevelup_pybind as mod
    md = synth.synth_model_depth(21)
    sd = _perturbed_scene(md, K_CAM.astype(np.float64), 2.0, (2.5, -2.0, 4.0), 21)
    ys, xs = np.nonzero(md)
    dx, dy = int(xs.min()), int(ys.min())
    R, t = np.eye(3, dtype=np.float32), np.array([0, 0, 1000], np.float32)
    for intended in (True, False):
        ctx = mod.IcpContext(device=0, scene_from_scene=intended)
        ctx.set_scene(sd, K_CAM)
        ctx.set_models([md])
        res, ms = ctx.run(K_CAM.reshape(1, 9), R.reshape(1, 9), t.reshape(1, 3), [(dx, dy)])
        ref = lo.pose_refine(sd, md, K_CAM, K_CAM, R, t, dx, dy, scene_from_scene=intended)
        src, tgt, nrm, dbg = (ctx.read_debug(0, k) for k in range(4))
        assert src.shape == ref["src"].shape and tgt.shape == ref["tgt"].shape
        assert np.abs(src - ref["src"]).max() < 1e-12 and np.abs(tgt - ref["tgt"]).max() < 1e-12
        assert np.abs(dbg[:3] - ref["init_guess"][:3, 3]).max() < 1e-12
        cosang = np.abs((nrm * ref["normals"]).sum(1))
        assert cosang.min() > 1 - 1e-9, cosang.min()
        assert res[0]["iterations"] == ref["iterations"] and abs(res[0]["residual"] - ref["residual"]) < 1e-6
        assert np.abs(res[0]["R"] - ref["R"]).max() < 1e-4 and np.abs(res[0]["t"] - ref["t"]).max() / 1000.0 < 1e-4
        ctx.close()


def test_icp_normals_of_a_cloud_with_depth_outliers(lm):
    """The kNN normals of a target cloud that carries what a depth sensor leaves behind: flying pixels a few centimetres off
    the surface (their ring has to grow: whole waves), single pixels and a blob of fewer than 30 points decimetres away (their
    30 nearest neighbours are the far side of the cloud: k_icp_knn_far, the whole cloud per point).  Neighbour sets are exact,
    so the normals agree with the oracle's brute force to 1e-9 (up to sign), and so does the registration."""
    import linemodLevelup_pybind as mod
    md = synth.synth_model_depth(33)
    sd = _perturbed_scene(md, K_CAM.astype(np.float64), 1.5, (2.0, 1.0, -3.0), 33)
    ys, xs = np.nonzero(md)
    dx, dy = int(xs.min()), int(ys.min())
    rng = np.random.default_rng(5)
    sd = sd.copy()
    inside = np.argwhere(sd > 0)
    pick = inside[rng.choice(len(inside), 40, replace=False)]
    for (y, x), off in zip(pick[:20], rng.integers(15, 60, 20)):          # flying pixels
        sd[y, x] = sd[y, x] + off
    for (y, x), off in zip(pick[20:28], (150, -120, 300, 420, -200, 250, 600, -90)):   # lone far pixels
        sd[y, x] = max(1, int(sd[y, x]) + off)
    y0, x0 = pick[30]
    sd[y0:y0 + 3, x0:x0 + 4] = np.where(sd[y0:y0 + 3, x0:x0 + 4] > 0, sd[y0:y0 + 3, x0:x0 + 4] + 800, 0)   # a blob of <= 12 points 0.8 m behind
    R, t = np.eye(3, dtype=np.float32), np.array([0, 0, 1000], np.float32)
    ctx = mod.IcpContext(device=0, scene_from_scene=True)
    ctx.set_scene(sd, K_CAM)
    ctx.set_models([md])
    res, ms = ctx.run(K_CAM.reshape(1, 9), R.reshape(1, 9), t.reshape(1, 3), [(dx, dy)])
    ref = lo.pose_refine(sd, md, K_CAM, K_CAM, R, t, dx, dy, scene_from_scene=True)
    tgt, nrm = ctx.read_debug(0, 1), ctx.read_debug(0, 2)
    assert tgt.shape == ref["tgt"].shape and np.abs(tgt - ref["tgt"]).max() < 1e-12
    assert (tgt[:, 2].max() - np.median(tgt[:, 2])) > 0.5                  # the far blob is in the cloud
    cosang = np.abs((nrm * ref["normals"]).sum(1))
    assert cosang.min() > 1 - 1e-9, (cosang.min(), int(np.argmin(cosang)))
    assert res[0]["iterations"] == ref["iterations"] and abs(res[0]["residual"] - ref["residual"]) < 1e-6
    assert np.abs(res[0]["R"] - ref["R"]).max() < 1e-4 and np.abs(res[0]["t"] - ref["t"]).max() / 1000.0 < 1e-4
    ctx.close()


@pytest.mark.parametrize("half_w,half_h", [(110, 100), (80, 75), (150, 120)])
def test_icp_large_clouds_take_the_global_sort_path(lm, half_w, half_h):
    """> 16k points per cloud: eight workgroups sort a cloud of up to 8 x 8192 points between them (159 x 149 = 23.7k and 219 x 199 =
    43.6k pixels); beyond that (299 x 239 = 71k) one workgroup does, its keys in HBM (bitonic network through the scratch) — and the
    team of 64 workgroups then holds more than 704 source points each, which takes the second team launch (two points per thread).
    The down-sampled cloud must still equal the oracle's, and registering the cloud to itself (verbatim mode, LL.cpp:109) is
    the identity."""
    import linemodLevelup_pybind as mod
    H, W = 480, 640
    yy, xx = np.mgrid[0:H, 0:W]
    inside = (np.abs(xx - W // 2) < half_w) & (np.abs(yy - H // 2) < half_h)
    md = np.where(inside, 2000 + ((xx - W // 2) * 0.5).astype(np.int64) + ((yy % 7) == 0) * 3, 0).astype(np.uint16)
    ys, xs = np.nonzero(md)
    dx, dy = int(xs.min()), int(ys.min())
    bp = lo.backproject_clouds(md, md, K_CAM, K_CAM, dx, dy)
    want = lo.voxel_down_sample(bp[0])
    assert len(bp[0]) > 16384 and len(want) > 16384
    ctx = mod.IcpContext(device=0, scene_from_scene=False)
    ctx.set_scene(md, K_CAM)
    ctx.set_models([md])
    res, ms = ctx.run(K_CAM.reshape(1, 9), np.eye(3, dtype=np.float32).reshape(1, 9), np.array([[0, 0, 2000]], np.float32), [(dx, dy)])
    src = ctx.read_debug(0, 0)
    assert src.shape == want.shape and np.abs(src - want).max() < 1e-12
    dbg = ctx.read_debug(0, 3)
    assert np.abs(dbg[:3] - bp[2]).max() < 1e-12
    assert res[0]["residual"] == 1.0 and res[0]["n_source"] == len(want) and res[0]["n_target"] == len(want)
    T = dbg[3:19].reshape(4, 4)
    assert np.abs(T[:3, :3] - np.eye(3)).max() < 1e-6 and np.abs(T[:3, 3]).max() < 1e-4
    ctx.close()


_ICP_PATHS_SCRIPT = r"""
import sys, json, os
import numpy as np
root = sys.argv[1]
sys.path[:0] = [root, os.path.join(root, "6dpose_amd"), os.path.join(root, "tests")]
import synth, linemodLevelup_pybind as mod
K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
rng = np.random.default_rng(3)
base = synth.synth_model_depth(41)
scene = np.where(base > 0, base + 3, 0).astype(np.uint16)
scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
mds, xy = [], []
for h in range(6):
    md = synth.synth_model_depth(41 + h % 3)
    ys, xs = np.nonzero(md)
    mds.append(md); xy.append((int(xs.min()) + h % 2, int(ys.min()) - h % 2))
n = len(mds)
Ks = np.tile(K.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1)); ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
ctx = mod.IcpContext(device=0, scene_from_scene=True)
ctx.set_scene(scene, K); ctx.set_models(mds)
res, _ = ctx.run(Ks, Rs, ts, xy)
res2, _ = ctx.run(Ks, Rs, ts, xy)
out = {"R": [r["R"].tolist() for r in res], "t": [r["t"].tolist() for r in res], "it": [int(r["iterations"]) for r in res],
       "again": all(np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) for a, b in zip(res, res2)),
       "src": [ctx.read_debug(h, 0).tolist() for h in range(2)], "tgt": [ctx.read_debug(h, 1).tolist() for h in range(2)]}
ctx.close()
print("RESULT " + json.dumps(out))
"""


def test_icp_preparation_paths_agree(lm, tmp_path):
    """The round-6 preparation (voxel down-sampling and search grid by eight workgroups per cloud, one point launch, boxes at upload) against the
    kernels of rounds 2-5 (`LM_ICP_WIDE_SORT=0`: one workgroup per cloud, two point launches, k_icp_bbox per run), each in its own process (the
    knob is read once): the down-sampled clouds are equal bit for bit, the poses to rounding (the search grid of the new path spans the
    extent of the cloud the target was down-sampled from, so the neighbour sums of the normals add up in another order), the iteration
    counts are equal, and a run repeats itself bit for bit on either path."""
    import subprocess, sys, json
    script = tmp_path / "icp_paths.py"
    script.write_text(_ICP_PATHS_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for wide in ("1", "0"):
        env = dict(os.environ, LM_ICP_WIDE_SORT=wide)
        r = subprocess.run([sys.executable, str(script), root], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
        text = r.stdout.decode()
        assert r.returncode == 0, text[-2000:]
        outs.append(json.loads([l for l in text.splitlines() if l.startswith("RESULT ")][-1][7:]))
    a, b = outs
    assert a["again"] and b["again"]
    assert a["it"] == b["it"] and max(a["it"]) > 5
    for h in range(2):
        assert np.array_equal(np.array(a["src"][h]), np.array(b["src"][h])) and np.array_equal(np.array(a["tgt"][h]), np.array(b["tgt"][h]))
    assert np.abs(np.array(a["R"]) - np.array(b["R"])).max() < 1e-9 and np.abs(np.array(a["t"]) - np.array(b["t"])).max() < 1e-6


@pytest.mark.parametrize("n", [72, 40])
def test_icp_more_hypotheses_than_the_kernel_deals_out(lm, n):
    """72 hypotheses (more than 64): k_icp_team runs on a fixed grid (a team of cus / count workgroups per hypothesis, here three) instead of
    dealing the workgroups out by cloud size, and its builds with two and five source points per thread take the hypotheses a workgroup of
    which holds more than 704 points.  40 hypotheses of ~2k points: a CRAMPED batch (its clouds could use 40 x 17 workgroups, the chip has
    256), which leaves the first launch after evaluation 3; the second launch deals the chip out among the hypotheses still at work and goes
    on from the state they left in IcpState.  Either way the poses must equal those of the same hypotheses run in batches of eight (teams
    of 16+, one launch), and a run must repeat itself bit for bit."""
    import linemodLevelup_pybind as mod
    base = synth.synth_model_depth(50)
    scene = _perturbed_scene(base, K_CAM.astype(np.float64), 1.0, (1.5, -1.0, 2.0), 50)
    mds = [synth.synth_model_depth(50 + s) for s in range(4)]
    slots = [h % 4 for h in range(n)]
    xy = []
    for h in range(n):
        ys, xs = np.nonzero(mds[slots[h]])
        xy.append((int(xs.min()) + (h % 5) - 2, int(ys.min()) + (h % 3) - 1))
    Ks = np.tile(K_CAM.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1))
    ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
    ctx = mod.IcpContext(device=0, scene_from_scene=True)
    ctx.set_scene(scene, K_CAM)
    ctx.set_models(mds)
    got, _ = ctx.run(Ks, Rs, ts, xy, model_slots=slots)
    again, _ = ctx.run(Ks, Rs, ts, xy, model_slots=slots)
    assert all(np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) and a["iterations"] == b["iterations"] for a, b in zip(got, again))
    assert max(g["iterations"] for g in got) > 6                  # (some hypotheses go on well beyond the cut)
    for b in range(0, n, 8):
        want, _ = ctx.run(Ks[b:b + 8], Rs[b:b + 8], ts[b:b + 8], xy[b:b + 8], model_slots=slots[b:b + 8])
        for g, w in zip(got[b:b + 8], want):
            assert g["iterations"] == w["iterations"] and g["n_source"] == w["n_source"] and g["n_target"] == w["n_target"]
            assert np.abs(g["R"] - w["R"]).max() < 1e-10 and np.abs(g["t"] - w["t"]).max() < 1e-7
    ctx.close()


def test_icp_context_slots_shared_by_hypotheses(lm):
    """Hypotheses may share a resident model slot; results equal the one-image-per-hypothesis batch call."""
    import linemodLevelup_pybind as mod
    base = synth.synth_model_depth(30)
    scene = _perturbed_scene(base, K_CAM.astype(np.float64), -1.0, (1.0, 2.0, 3.0), 30)
    mds = [synth.synth_model_depth(30 + s) for s in range(3)]
    slots = [0, 2, 1, 2, 0]
    xy = []
    for sl in slots:
        ys, xs = np.nonzero(mds[sl])
        xy.append((int(xs.min()) + len(xy) - 2, int(ys.min()) + 1))
    n = len(slots)
    Ks = np.tile(K_CAM.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1))
    ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
    ctx = mod.IcpContext(device=0, scene_from_scene=True)
    ctx.set_scene(scene, K_CAM)
    ctx.set_models(mds)
    got, _ = ctx.run(Ks, Rs, ts, xy, model_slots=slots)
    want, _ = mod.pose_refine_batch(scene, K_CAM, [mds[sl] for sl in slots], Ks, Rs, ts, xy, device=0, scene_from_scene=True)
    for g, w in zip(got, want):
        assert np.array_equal(g["R"], w["R"]) and np.array_equal(g["t"], w["t"]) and g["iterations"] == w["iterations"]
    again, _ = ctx.run(Ks, Rs, ts, xy, model_slots=slots)                                     # deterministic run to run
    for g, w in zip(got, again):
        assert np.array_equal(g["R"], w["R"]) and np.array_equal(g["t"], w["t"])
    ctx.close()


# ---------------------------------------------------------------------------------------------
# per-frame pipeline: match -> boxes -> nms -> top-k -> poseRefine, all on the device
# ---------------------------------------------------------------------------------------------
def _pipeline_reference(mod, det, rgb, dep, wh, E, views, thr, top_k, iou, box=None):
    """The driver loop (linemod_and_levelup_test.py:324-372) with the product's reference-shaped calls."""
    m = det.matchArray([rgb, dep], thr, ["obj"])
    dets = np.zeros((len(m), 5))
    for i, r in enumerate(m):
        w, h = wh[int(r["template_id"]) * E] if box is None else box[int(r["template_id"])]
        dets[i] = (r["x"], r["y"], r["x"] + w, r["y"] + h, r["similarity"])
    keep = mod.nms(dets, iou)[:top_k]
    sel = [m[i] for i in keep]
    mds = [views[int(r["template_id"])][0] for r in sel]
    Ks = np.stack([views[int(r["template_id"])][1] for r in sel]); Rs = np.stack([views[int(r["template_id"])][2] for r in sel])
    ts = np.stack([views[int(r["template_id"])][3] for r in sel])
    xy = [(int(r["x"]), int(r["y"])) for r in sel]
    poses, _ = mod.pose_refine_batch(dep, K_CAM, mds, Ks, Rs, ts, xy, device=0, scene_from_scene=True)
    return sel, poses, len(m)


def _pipeline_oracle(od, rgb, dep, bank, T, wh, E, views, thr, top_k, iou, box=None):
    """The same driver loop on the CPU ORACLE only (nothing of the product): match_oracle.c -> canonical sort/unique ->
    numpy nms (the driver's own function) -> oracle poseRefine per kept match."""
    raw, _ = oracle_matches(od, rgb, dep, bank, T, thr)
    m = lo.canonical_sort_unique(raw)
    dets = np.zeros((len(m), 5))
    for i, r in enumerate(m):
        w, h = wh[int(r["tid"]) * E] if box is None else box[int(r["tid"])]
        dets[i] = (r["x"], r["y"], r["x"] + w, r["y"] + h, r["sim"])
    keep = lo.nms_boxes(dets, iou, stable=True)[:top_k] if len(m) else []      # planted templates tie in score
    sel = [m[i] for i in keep]
    poses = []
    for r in sel:
        md, K, R, t = views[int(r["tid"])]
        poses.append(lo.pose_refine(dep, md, K_CAM, K, R, t, int(r["x"]), int(r["y"]), scene_from_scene=True))
    return sel, poses


@pytest.mark.parametrize("dup", [False, True])
def test_pipeline_equals_match_nms_pose_refine(lm, dup):
    """lm_pipeline_run on the device = Detector.match + nms + poseRefine of the reference driver, detection by
    detection.  dup=True appends copies of templates under new ids: equal (x, y, similarity) across template ids
    exercises the adjacent-unique rule of Detector::match inside the on-device NMS."""
    W, H, T, nfeat, n = 640, 480, [4, 8], (64, 32), 60
    rgb, dep = synth.make_frame(11, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    feat, offs, wh = synth.make_planted_bank(77, n, [(p[0], p[1]) for p in pyr], T, nfeat)
    E = 2 * len(T)
    if dup:                                                     # templates 0..19 again as ids n..n+19
        extra_f = feat[:offs[20 * E]]
        extra_o = offs[1:20 * E + 1] + offs[-1]
        feat = np.concatenate([feat, extra_f]); offs = np.concatenate([offs, extra_o]).astype(np.int32); wh = np.concatenate([wh, wh[:20 * E]])
        n += 20
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("obj", feat, offs, wh)
    rng = np.random.default_rng(5)
    shapes = [synth.synth_model_depth(200 + k, W, H) for k in range(4)]
    views = []
    for t in range(n):
        R = np.eye(3, dtype=np.float32)
        tt = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 1000 + rng.uniform(-20, 20)], np.float32)
        views.append((shapes[t % 4], K_CAM.copy(), R, tt))
    pipe = lm.Pipeline(det, W, H, scene_from_scene=True)
    # dup=True also overrides the NMS boxes with the driver's aTemplateInfo width / height (the rendered depth's extent)
    box = None if not dup else [(int(wh[t * E][0]) + 7 - t % 5, int(wh[t * E][1]) - 4 + t % 3) for t in range(n)]
    pipe.set_views("obj", [v[0] for v in views], [v[1] for v in views], [v[2] for v in views], [v[3] for v in views], box_wh=box)
    for thr, top_k in ((70.0, 8), (85.0, 5)):
        sel, poses, n_matches = _pipeline_reference(lm, det, rgb, dep, wh, E, views, thr, top_k, 0.5, box)
        det.setFrame([rgb, dep])
        got, tm = pipe.run(thr, ["obj"], K_CAM, top_k=top_k, nms_iou=0.5)
        assert len(got) == len(sel) and len(got) > 0
        for g, r, p in zip(got, sel, poses):
            assert (g["x"], g["y"], g["template_id"]) == (int(r["x"]), int(r["y"]), int(r["template_id"]))
            assert g["similarity"] == float(r["similarity"])
            assert (g["width"], g["height"]) == (tuple(int(v) for v in wh[int(r["template_id"]) * E]) if box is None else box[int(r["template_id"])])
            if p["residual"] == -1.0:
                assert g["status"] == 1 and g["residual"] == -1.0
                continue
            assert g["status"] == 0 and g["iterations"] == p["iterations"] and abs(g["residual"] - p["residual"]) < 1e-6
            if len(got) == top_k:                                # same hypothesis count = same slicing: identical sums
                assert np.allclose(g["R"], p["R"], atol=1e-9, equal_nan=True) and np.allclose(g["t"], p["t"], atol=1e-6, equal_nan=True)
            else:
                assert np.allclose(g["R"], p["R"], atol=1e-6, equal_nan=True) and np.allclose(g["t"], p["t"], atol=1e-3, equal_nan=True)
        assert tm["total_ms"] > 0 and tm["coarse_candidates"] > 0
        # the ROS node's translation NMS on top (linemod_ros/detect.py:128): same survivors as numpy nms_norms on the results
        posed = [g for g in got if g["status"] == 0]
        if len(posed) > 1:
            keep = lo.nms_norms(np.array([g["t"] for g in posed]).reshape(-1, 3), np.array([-g["residual"] for g in posed]), 40.0)
            det.setFrame([rgb, dep])
            got_n, _ = pipe.run(thr, ["obj"], K_CAM, top_k=top_k, nms_iou=0.5, norms_thresh=40.0)
            if len(set(g["residual"] for g in posed)) == len(posed):      # tied residuals: numpy's argsort decides the order
                assert [(g["x"], g["y"], g["template_id"]) for g in got_n] == [(posed[i]["x"], posed[i]["y"], posed[i]["template_id"]) for i in keep]
            assert 1 <= len(got_n) <= len(posed)
        # ... and against the ORACLE's own chain (no product call on the expected side): detections exact, poses to 1e-4
        osel, oposes = _pipeline_oracle(od, rgb, dep, (feat, offs, wh), T, wh, E, views, thr, top_k, 0.5, box)
        assert len(got) == len(osel)
        for g, r, p in zip(got, osel, oposes):
            assert (g["x"], g["y"], g["template_id"], g["similarity"]) == (int(r["x"]), int(r["y"]), int(r["tid"]), float(r["sim"]))
            if p["residual"] == -1.0:
                assert g["status"] == 1
                continue
            assert g["status"] == 0 and abs(g["residual"] - p["residual"]) < 1e-6
            assert np.allclose(g["R"], p["R"], atol=1e-4) and np.allclose(np.ravel(g["t"]) / 1000.0, np.ravel(p["t"]) / 1000.0, atol=1e-4)
    pipe.close()


# ---------------------------------------------------------------------------------------------
# multi-process sharding on the one visible GPU
# ---------------------------------------------------------------------------------------------
_SHARD_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
root, port, rank, world, backend = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sys.path[:0] = [os.path.join(root, "6dpose_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")]
import linemodLevelup_pybind as lm, sharded, synth, linemod_oracle as lo
kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
dist.init_process_group(backend, init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world, **kw)
W, H, T, nfeat = 640, 480, [4, 8], (150, 75)
rgb, dep = synth.make_frame(13, W, H)
od = lo.OracleDetector(nfeat[0], T)
pyr = od.quantize_pyramid(rgb, dep)
banks = {c: synth.make_planted_bank(41 + i, 90 + 20 * i, [(p[0], p[1]) for p in pyr], T, nfeat) for i, c in enumerate(["a", "b", "c"])}
det = lm.Detector(nfeat[0], T, device=0)
for c in banks:
    det.addClassPacked(c, *banks[c])
dev = "cuda:0" if backend == "nccl" else None
got = sharded.match_sharded(det, [rgb, dep], 75.0, ["c", "a", "b"], device=dev)
if backend == "nccl":   # single rank: also push the records through the RCCL all-gather explicitly
    det.setFrame([rgb, dep])
    pre = det.matchResident(75.0, ["c", "a", "b"], sort_unique=False)
    again = lm.merge_matches(sharded.gather_records(pre, device=dev, force=True))
    assert again.tobytes() == got.tobytes()
# the same exchange as device work: sort per rank, all-gather of the blocks (RCCL / staged for gloo), ranking merge
ex = sharded.DeviceExchange(det, "cuda:0", force=True)
dev_got = sharded.match_sharded(det, [rgb, dep], 75.0, ["c", "a", "b"], device=dev, exchange=ex)
assert dev_got.tobytes() == got.tobytes(), (rank, len(dev_got), len(got))
if backend == "nccl":   # the all-gather above was issued by the C library (lm_exchange_allgather, its own RCCL communicator); and once through torch's
    assert ex.comm is not None and lm.Comm.available()
    os.environ["LM_EXCHANGE_COLLECTIVE"] = "torch"
    ex_t = sharded.DeviceExchange(det, "cuda:0", force=True)
    assert ex_t.comm is None
    assert sharded.match_sharded(det, [rgb, dep], 75.0, ["c", "a", "b"], device=dev, exchange=ex_t).tobytes() == got.tobytes()
    del os.environ["LM_EXCHANGE_COLLECTIVE"]
frames = [synth.make_frame(13 + k, W, H) for k in range(4)]
for k, f in enumerate(frames):
    det.storeFrame(k, f)
want = []
for k in range(4):
    det.selectFrame(k)
    want.append(sharded.match_sharded(det, None, 75.0, ["c", "a", "b"], device=dev, resident=True))
outs = []
for k in range(6):                                          # three frames in flight
    det.selectFrame(k % 4); ex.submit(75.0, ["c", "a", "b"])
    if k >= 2:
        outs.append(ex.collect())
outs += [ex.collect(), ex.collect()]
for k, o in enumerate(outs):
    assert o is not None and o.tobytes() == want[k % 4].tobytes(), (rank, k)
small = sharded.DeviceExchange(det, "cuda:0", capacity=256, force=True)      # a block too small: every rank falls back together
det.selectFrame(0)
fb = sharded.match_sharded(det, None, 60.0, ["c", "a", "b"], device=dev, resident=True, exchange=small)
assert fb.tobytes() == sharded.match_sharded(det, None, 60.0, ["c", "a", "b"], device=dev, resident=True).tobytes()
assert small.capacity > 256 or world > 1
det.setShard(0, 1)
whole = det.matchArray([rgb, dep], 75.0, ["c", "a", "b"])
assert len(whole) > 0 and got.tobytes() == whole.tobytes(), (rank, len(got), len(whole))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", len(got))
'''


def _run_workers(tmp_path, world, backend):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER)
    port = str(29800 + os.getpid() % 1500)
    procs = [subprocess.Popen([sys.executable, str(script), root, port, str(r), str(world), backend],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=500)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


def test_match_sharded_two_processes_one_gpu_gloo(lm, tmp_path):
    """world_size 2, both ranks computing on cuda:0, records exchanged over gloo: equals the unsharded result."""
    _run_workers(tmp_path, 2, "gloo")


def test_cpp_caller_exchanges_through_the_librarys_own_rccl_collective(lm, tmp_path):
    """tests/cpp/exchange_rccl_smoke.cpp — a C++ program shaped like the reference's test.cpp (train, match per frame), no Python: frames go
    submit -> lm_detector_exchange_group (pack + ncclAllGather issued by the library on the exchange stream + merge) -> collect and must
    equal lm_detector_match record by record.  World size 1 (one GPU here); the same binary is one rank of N."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "exchange_rccl_smoke")
    subprocess.check_call([hipcc, "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "exchange_rccl_smoke.cpp"),
                           "-L", os.path.join(root, "6dpose_amd"), "-lamdlinemod", "-Wl,-rpath," + os.path.join(root, "6dpose_amd"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"ok:" in r.stdout, r.stdout.decode()[-2000:]


def test_match_sharded_rccl_single_rank(lm, tmp_path):
    """The RCCL (backend "nccl") all-gather path with the one rank a 1-GPU box allows."""
    _run_workers(tmp_path, 1, "nccl")


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launcher_runs_two_ranks(lm, scaling):
    """`python bench.py --gpus 2` (no torchrun around it) spawns two ranks — here both on the one GPU with the gloo
    backend — and rank 0 prints ONE JSON line with n_gpus = 2, the live-stream frame source and the device exchange."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LM_BENCH_BACKEND="gloo", LM_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--templates", "150", "--scaling", scaling, "--no-extras", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks_observed"] == 2 and j["scaling"] == scaling
    assert j["config"]["templates_total"] == (1200 if scaling == "strong" else 300)
    assert "host memory" in j["config"]["frame_source"] and j["stages_ms"]["h2d_ms"] > 0
    assert j["config"]["exchange"] == "device" and j["value"] > 0 and j["config"]["matches_final_last_step"] > 0
