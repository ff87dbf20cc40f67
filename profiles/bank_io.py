"""Load / store times of a template bank: the reference's one-YAML-per-class files (writeClasses / readClasses, LL.cpp:2124-2146)
against the packed binary bank (writeBank / readBank, csrc/bank_file.cpp), at 2k templates (both) and at BASELINE configs[4]'s
30 objects x 3k views (packed only; the YAML time is the 2k figure scaled)."""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, synth

def timed(f):
    t0 = time.perf_counter(); f(); return time.perf_counter() - t0

res = {}
tmp = tempfile.mkdtemp()
bank = synth.make_random_bank(5, 2000, 640, 480, (150, 75))
a = lm.Detector(150, [4, 8], device=0)
a.addClassPacked("obj", *bank)
res["yaml_write_s_2k"] = timed(lambda: a.writeClasses(os.path.join(tmp, "%s.yaml")))
res["yaml_bytes_2k"] = os.path.getsize(os.path.join(tmp, "obj.yaml"))
b = lm.Detector(150, [4, 8], device=0)
res["yaml_read_s_2k"] = timed(lambda: b.readClasses(["obj"], os.path.join(tmp, "%s.yaml")))
res["packed_write_s_2k"] = timed(lambda: a.writeBank(os.path.join(tmp, "b2k.lmb")))
res["packed_bytes_2k"] = os.path.getsize(os.path.join(tmp, "b2k.lmb"))
c = lm.Detector(150, [4, 8], device=0)
res["packed_read_s_2k"] = timed(lambda: c.readBank(os.path.join(tmp, "b2k.lmb")))
assert all(np.array_equal(x.features, y.features) for t in (0, 999, 1999) for x, y in zip(b.getTemplates("obj", t), c.getTemplates("obj", t)))
# 30 objects x 3000 views
big = lm.Detector(150, [4, 8], device=0)
b3k = synth.make_random_bank(6, 3000, 640, 480, (150, 75))
for o in range(30):
    big.addClassPacked("obj_%02d" % o, *b3k)
res["packed_write_s_90k"] = timed(lambda: big.writeBank(os.path.join(tmp, "b90k.lmb")))
res["packed_bytes_90k"] = os.path.getsize(os.path.join(tmp, "b90k.lmb"))
d = lm.Detector(150, [4, 8], device=0)
res["packed_read_s_90k"] = timed(lambda: d.readBank(os.path.join(tmp, "b90k.lmb")))
e = lm.Detector(150, [4, 8], device=0)
res["packed_read_s_one_object_of_30"] = timed(lambda: e.readBank(os.path.join(tmp, "b90k.lmb"), ["obj_17"]))
assert d.numTemplates() == 90000 and e.numTemplates() == 3000
res["yaml_read_s_90k_scaled"] = res["yaml_read_s_2k"] * 45
print(json.dumps(res))
