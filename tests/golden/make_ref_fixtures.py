"""Writes tests/golden/ref_expected.json from oracle/_ref — the reference's OWN match lines
(LL.cpp:1022-1658, 1694-1941, compiled from /root/reference against oracle/ref_shim; `make -C oracle`).
Run in the build container (needs /root/reference to build _ref); the JSON is committed so that the
pin also holds where oracle/_ref is absent.

Per case of tests/ref_cases.py: sha1 of the linear memories of every level/modality, and per
(threshold, class_ids) the reference's pre-unique match list (sha1 of the records in the order the
reference appends them + count) and the output of Detector::match itself (count after its
std::sort + std::unique, sha1 of the records sorted by all fields)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ll_ref            # noqa: E402
import ref_cases as rc   # noqa: E402


def main():
    assert ll_ref.available("sse2") and ll_ref.available("ssse3"), "build oracle/_ref first: make -C oracle"
    out = {}
    for case in rc.all_cases():
        q = rc.quantized_of(case)
        T = case["T"]
        e = {"lm": [[rc.sha(ll_ref.build_linear_memories(q[l][m], T[l])) for m in range(2)] for l in range(len(T))],
             "match": {}}
        for thr in case["thresholds"]:
            for req in case["requests"]:
                pre = ll_ref.match(q, T, case["banks"], thr, req, pre_unique=True)
                fin = ll_ref.match(q, T, case["banks"], thr, req, pre_unique=False)
                for v in ("ssse3",):
                    assert ll_ref.match(q, T, case["banks"], thr, req, pre_unique=True, variant=v).tolist() == pre.tolist()
                e["match"][rc.record_key(thr, req)] = {
                    "pre_unique_n": len(pre), "pre_unique_sha1": rc.sha(pre),
                    "final_n": len(fin),
                    "final_distinct_n": len(set(zip(fin["x"].tolist(), fin["y"].tolist(), fin["sim"].tolist(), fin["cls"].tolist()))), "final_sorted_sha1": rc.sha(np.sort(fin, order=["x", "y", "sim", "cls", "tid"])),
                    "top": [int(fin[0]["x"]), int(fin[0]["y"]), float(fin[0]["sim"]).hex(), int(fin[0]["tid"])] if len(fin) else None,
                }
        out[case["name"]] = e
        print(case["name"], {k: (v["pre_unique_n"], v["final_n"]) for k, v in e["match"].items()})
    with open(os.path.join(HERE, "ref_expected.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
