"""Host cost of the device exchange on the calling thread (world 1): live-stream loop with sharded.DeviceExchange, with and without
the torch collective.  GPU box: python profiles/exchange_host_profile.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench, linemodLevelup_pybind as lm, synth, sharded
W, H = bench.W, bench.H
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if os.environ.get("PG_FIRST"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
det = lm.Detector(bench.NFEAT[0], bench.T_LEVELS, device=0)
frames = bench.noisy_frames(16)
det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
det.setFrame(list(frames[0])); det.matchResident(75.0, ["_probe"])
quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
det.addClassPacked("obj", *synth.make_planted_bank(1234, 2000, quant, bench.T_LEVELS, bench.NFEAT))
cls = ["obj"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for mode in os.environ.get("MODES", "plain,exchange_no_collective,exchange_rccl").split(","):
    ex = None
    if mode == "exchange_rccl" and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if mode != "plain":
        ex = sharded.DeviceExchange(det, dev, force=(mode == "exchange_rccl"), shard=False)
        if os.environ.get("CHECK_FIRST"):
            det.setFrame(list(frames[0]))
            a = sharded.match_sharded(det, None, 75.0, cls, device=dev, resident=True, shard=False)
            b = sharded.match_sharded(det, None, 75.0, cls, device=dev, resident=True, exchange=ex, shard=False)
            assert a.tobytes() == b.tobytes()
    buf = np.empty(ex.capacity if ex else 1, lm.MATCH_DTYPE)
    T = {"submit": 0.0, "pump": 0.0, "collect": 0.0}
    def run(n, depth=12):
        infl = 0
        for k in range(n):
            t0 = time.perf_counter()
            if ex is None: det.submitFrame(frames[k % 16], 75.0, cls)
            else: ex.submit(75.0, cls, frame=frames[k % 16])
            T["submit"] += time.perf_counter() - t0; infl += 1
            if infl == depth:
                t0 = time.perf_counter()
                (det.collect() if ex is None else ex.collect(into=buf)); T["collect"] += time.perf_counter() - t0; infl -= 1
        while infl:
            t0 = time.perf_counter()
            (det.collect() if ex is None else ex.collect(into=buf)); T["collect"] += time.perf_counter() - t0; infl -= 1
    run(48)
    for q in T: T[q] = 0.0
    t0 = time.perf_counter(); run(400); dt = time.perf_counter() - t0
    print("%-24s %.4f ms/frame   submit %.1f us  collect %.1f us" % (mode, dt / 400 * 1e3, T["submit"] / 400 * 1e6, T["collect"] / 400 * 1e6), flush=True)
    if mode != "plain":
        import cProfile, pstats, io
        pr = cProfile.Profile(); pr.enable(); run(200); pr.disable()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(12); print(st.getvalue()[:2600], flush=True)
    det.setAsyncCollect(False)
