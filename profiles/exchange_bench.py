"""The device exchange at 8 ranks, emulated on one GPU: 8 detectors hold the BASELINE configs[3] bank (8 objects x 2000 templates),
detector r searches shard r of 8, their packed blocks are laid out as an all-gather would and ONE of them merges — the work a
rank does per frame next to its matching kernels.  Reports the host time of the exchange calls and (through torch events on the
exchange stream) the device time of sort + merge + copy; run under rocprofv3 --kernel-trace for the per-kernel split.
Also times the host path (gather through host memory is not emulated: only the merge sort every rank would do)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import torch
import linemodLevelup_pybind as lm, synth
W, H, T, NF, NT, WORLD, CAP, THR = 640, 480, [4, 8], (150, 75), 2000, 8, 4096, 75.0
rgb, dep = synth.make_frame(0, W, H)
probe = lm.Detector(NF[0], T, device=0)
probe.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
probe.setFrame([rgb, dep]); probe.matchResident(THR, ["_probe"])
quant = [(probe.readStage(l, 0).reshape(H >> l, W >> l), probe.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
banks = [synth.make_planted_bank(1234 + o, NT, quant, T, NF) for o in range(WORLD)]
classes = ["obj%02d" % o for o in range(WORLD)]
dets = []
for r in range(WORLD):
    d = lm.Detector(NF[0], T, device=0)
    for c, b in zip(classes, banks):
        d.addClassPacked(c, *b)
    d.setFrame([rgb, dep]); d.setShard(r, WORLD)
    dets.append(d)
nb = lm.load_library().lm_exchange_block_bytes(CAP)
send = [torch.zeros(nb, dtype=torch.uint8, device="cuda:0") for _ in range(WORLD)]
streams = [torch.cuda.ExternalStream(d.exchangeStream(), device="cuda:0") for d in dets]
res = {"world": WORLD, "templates_per_rank": NT, "capacity": CAP}
pre = []
for r, d in enumerate(dets):
    pre.append(d.matchResident(THR, classes, sort_unique=False, distinct=True))
res["distinct_records_per_rank"] = [len(p) for p in pre]
t0 = time.perf_counter()
for _ in range(10):
    want = lm.merge_matches(np.concatenate(pre))
res["host_merge_ms_all_records"] = (time.perf_counter() - t0) / 10 * 1e3
res["final_matches"] = len(want)
REPS = 20
xbuf = np.empty(WORLD * CAP, lm.MATCH_DTYPE)
host_pack = host_merge = host_collect = 0.0
dev_ms = []
for it in range(REPS + 3):
    for r, d in enumerate(dets):
        d.submit(THR, classes)
        t0 = time.perf_counter(); d.exchangePack(send[r].data_ptr(), CAP); t1 = time.perf_counter()
        if r == 0 and it >= 3: host_pack += t1 - t0
    for st in streams[1:]:
        st.synchronize()
    recv = torch.cat(send)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(streams[0])
    t0 = time.perf_counter(); dets[0].exchangeMerge(recv.data_ptr(), WORLD, CAP); t1 = time.perf_counter()
    e1.record(streams[0])
    got, failed = dets[0].exchangeCollectInto(xbuf); t2 = time.perf_counter()
    assert failed == 0 and got.tobytes() == want.tobytes()
    if it >= 3:
        host_merge += t1 - t0; host_collect += t2 - t1; dev_ms.append(e0.elapsed_time(e1))
    for d in dets[1:]:
        d.collect(sort_unique=False)
res.update(host_pack_call_ms=host_pack / REPS * 1e3, host_merge_call_ms=host_merge / REPS * 1e3,
           host_collect_ms_incl_wait=host_collect / REPS * 1e3, device_merge_plus_copy_ms=float(np.median(dev_ms)))
print(json.dumps(res))
