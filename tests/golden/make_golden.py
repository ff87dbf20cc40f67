"""Regenerates tests/golden/ from the read-only reference checkout (run in the build container only;
/root/reference does not exist on the GPU box, so the outputs are committed).

Inputs copied verbatim (they are the reference's own test fixtures, SURVEY Appendix D — data, not
source): training images + golden template YAML (`test.cpp:36-51`), scene frame 0000, template
its half-occluded variant (test.cpp:95-96), template
banks 63/127/600 (gzipped), the poseRefine depth images.  `expected.json` records the stage hashes and
match lists of SURVEY Appendix C.2 (produced by an independent numpy restatement that reproduced
the golden YAML), which pin oracle/linemod_oracle.py + oracle/match_oracle.c.
"""
import gzip, hashlib, json, os, shutil, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/linemodLevelup/test/case1/"
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))


def main():
    for f in ["train_rgb.png", "train_dep.png", "train_mask.png", "0000_rgb.png", "0000_dep.png",
              "0000_rgb_half.png", "0000_dep_half.png"]:
        shutil.copyfile(REF + f, os.path.join(HERE, f))
    shutil.copyfile(REF + "pose/depth_ren.png", os.path.join(HERE, "pose_depth_ren.png"))
    shutil.copyfile(REF + "pose/0003.png", os.path.join(HERE, "pose_0003.png"))
    shutil.copyfile(REF + "writeClasses/06_template.yaml", os.path.join(HERE, "writeClasses_06_template.yaml"))
    for bank in ["63", "127", "600"]:
        with open(REF + bank + "/06_template.yaml", "rb") as fi, \
                gzip.GzipFile(os.path.join(HERE, "bank%s_06_template.yaml.gz" % bank), "wb", mtime=0) as fo:
            fo.write(fi.read())
    # SURVEY Appendix C.2 values (frame 0000 read as BGR, T={5,8})
    expected = {
        "normal_lut_sha1": "3ea8ffc4d964bbef5908b6609dc2bbeb551d1f26",
        "frame0000_bgr": {
            "pyrdown": "c297eac05cc8d693",
            "ori": ["f7d0e74137434c7c", "340c58941bbb4ffe"], "ori_nz": [148999, 37247],
            "nrm": ["1074b2f6996d199c", "08e2e4c0979982ec"], "nrm_nz": [280563, 69984],
            "spread_ori": ["61fb6c95a158c11c", "79724157e39b2e1e"],
            "spread_nrm": ["ba777836b583ab90", "1de717dc750d3e2e"],
            "lm_ori": ["49785b1b5b75b011", "d31ae326117fdfe8"],
            "lm_nrm": ["ba9cc001c1be5499", "1176731a068f2c67"],
        },
        "train_bgr": {
            "pyrdown": "8f29e84a30980d87",
            "ori": ["7af0fc94759ba412", "7a165a94ea1861c6"], "ori_nz": [2682, 1028],
            "nrm": ["97432d6a02540a55", "dbcf5ebfbeeee6d5"], "nrm_nz": [3058, 767],
        },
        "match_thr75": {
            "127": {"templates": 89, "coarse_candidates": 1184,
                    "pre_unique": [[332, 127, "0x1.4270e2p+6", 34]] * 3 + [[332, 132, "0x1.2f8b16p+6", 7]] * 4},
            "63": {"templates": 89, "coarse_candidates": 1214,
                   "pre_unique": [[332, 127, "0x1.3fd75ep+6", 34]] * 2},
        },
    }
    with open(os.path.join(HERE, "expected.json"), "w") as fh:
        json.dump(expected, fh, indent=1)
    print("golden written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
