"""Pins the CPU oracle's MATCH half (oracle/match_oracle.c: spread, response, linearise,
similarity*, matchClass — SURVEY rows A5-A12) to the reference's own lines.

oracle/_ref (oracle/Makefile) compiles LL.cpp:1022-1658 and 1694-1941 unmodified from
/root/reference against a cv::Mat buffer shim.  Two layers:
 (1) tests/golden/ref_expected.json — written from oracle/_ref by tests/golden/make_ref_fixtures.py
     and committed: checked always.
 (2) oracle/_ref itself, when the built library is present (build container; it also travels to
     the GPU box): record-by-record comparison, both SIMD builds, plus Detector::match's own
     std::sort/std::unique output against the repo's canonical order.
CPU only."""
import json
import os

import numpy as np
import pytest

import linemod_oracle as lo
import ll_ref
import ref_cases as rc
from helpers import GOLDEN

EXP = json.load(open(os.path.join(GOLDEN, "ref_expected.json")))
CASES = [("fixture", f, b) for f in ("", "_half") for b in ("63", "127", "600")] + [("synth", i, None) for i in range(len(rc.SYNTH))]


def _case(kind, a, b):
    return rc.fixture_case(a, b) if kind == "fixture" else rc.synth_case(a)


@pytest.mark.parametrize("kind,a,b", CASES)
def test_oracle_equals_reference_lines(kind, a, b):
    case = _case(kind, a, b)
    e = EXP[case["name"]]
    q, T = rc.quantized_of(case), case["T"]
    have_ref = ll_ref.available("sse2")
    for l in range(len(T)):
        for m in range(2):
            n = 8 * q[l][m].size
            lm = lo.build_linear_memories(q[l][m], T[l])[:n]
            assert rc.sha(lm) == e["lm"][l][m]
            if have_ref:
                for v in ll_ref.VARIANTS:
                    if ll_ref.available(v):
                        assert np.array_equal(lm, ll_ref.build_linear_memories(q[l][m], T[l], v))
    n_pre = 0
    for thr in case["thresholds"]:
        for req in case["requests"]:
            x = e["match"][rc.record_key(thr, req)]
            raw, _ = rc.oracle_run(case, thr, req)
            n_pre += len(raw)
            # same records in the same order as the reference appends them (template by template)
            assert (len(raw), rc.sha(raw)) == (x["pre_unique_n"], x["pre_unique_sha1"])
            # the repo's canonical sort + unique keeps the same distinct (x, y, sim, class)
            # as the reference's own std::sort + std::unique; the reference keeps MORE entries because
            # its sort key ignores x, y, so equal records are not always adjacent for std::unique
            can = lo.canonical_sort_unique(raw)
            assert len(set(zip(can["x"].tolist(), can["y"].tolist(), can["sim"].tolist(), can["cls"].tolist()))) == x["final_distinct_n"]
            assert len(can) <= x["final_n"] or x["final_n"] == 0
            if have_ref:
                for v in ll_ref.VARIANTS:
                    if not ll_ref.available(v):
                        continue
                    pre = ll_ref.match(q, T, case["banks"], thr, req, pre_unique=True, variant=v)
                    assert pre.tolist() == raw.tolist()
                    fin = ll_ref.match(q, T, case["banks"], thr, req, pre_unique=False, variant=v)
                    key = lambda r: set(zip(r["x"].tolist(), r["y"].tolist(), r["sim"].tolist(), r["cls"].tolist()))
                    assert key(fin) == key(can)
                    assert np.all(np.diff(fin["sim"]) <= 0)          # similarity is non-increasing
    assert n_pre > 0 or kind == "fixture"


def test_ref_builds_cover_both_simd_paths():
    """-O3 -Wall (the reference's flags) takes the SSE2 paths, -mssse3 the LDDQU/pshufb ones."""
    if not ll_ref.available("sse2"):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    assert ll_ref.lib("sse2").ref_simd_flags() == 1
    assert ll_ref.lib("ssse3").ref_simd_flags() == 7


def test_reference_sort_unique_keeps_duplicates_the_canonical_order_removes():
    """Documented difference: Match::operator< ignores x, y and operator== ignores
    template_id, and std::sort is unstable, so libstdc++ leaves equal records non-adjacent and
    std::unique keeps some of them: on fixture bank 63 at threshold 55 the reference returns 560
    entries holding 358 distinct (x, y, similarity); the canonical order returns those 358 plus the
    few that differ only by template id across a gap.  Positions and scores are identical."""
    if not ll_ref.available("sse2"):
        pytest.skip("oracle/_ref not built")
    case = rc.fixture_case("", "63")
    q = rc.quantized_of(case)
    raw, _ = rc.oracle_run(case, 55.0, ["06_template"])
    can = lo.canonical_sort_unique(raw)
    fin = ll_ref.match(q, case["T"], case["banks"], 55.0, ["06_template"])
    d = lambda r: set(zip(r["x"].tolist(), r["y"].tolist(), r["sim"].tolist()))
    assert (len(raw), len(fin), len(d(fin)), len(d(can))) == (4702, 560, 358, 358)
    assert 358 <= len(can) < len(fin)
