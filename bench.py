#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X linemodLevelup hot path (BASELINE.json metric:
templates·Mpixels matched/sec on 640x480 RGB-D).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of Detector.match's device path over one synthetic 640x480 RGB-D frame with
the template bank resident: front end (quantise, spread, response, linearise) + coarse similarity
over all templates + 16x16 refinement of every candidate + download of the match records (+, for
N>1, the all-gather of the per-rank records over RCCL and the canonical merge).  Frames are parked
in HBM before the timed region (lm_detector_store_frame) and made current with a device-to-device
copy, so `value` is the rate with inputs resident in HBM (DESIGN.md notes the PCIe-inclusive rate).

Workload (N=1): BASELINE configs[1] — 1 object x 2000 template pyramids, Detector(150,[4,8])
(150+150 features at level 0, 75+75 at level 1), threshold 75, planted synthetic bank (6dpose_amd/
synth.py).  N>1: configs[3] shape — N objects x 2000 templates, one object per rank (weak scaling),
every rank searches its contiguous slice of the bank and the match records are all-gathered.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel (k_local or k_coarse):
algorithmic response bytes per launch (sum nfeat*256 per 16x16 evaluation, resp.
sum nfeat*template_positions per template) / that kernel's mean duration from HIP events on the
detector's stream; peak = 8000 GB/s (HBM3E spec).  `cpu_baseline` times the oracle's SSE C port of
the same matching step on the host (rank 0, N=1 only), single thread like the reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd"))

W, H = 640, 480
T_LEVELS = [4, 8]
NFEAT = (150, 75)
N_TEMPLATES = 2000
THRESHOLD = 75.0
N_FRAMES = 4
PIPELINE_DEPTH = int(os.environ.get("LM_BENCH_DEPTH", "3"))   # frames in flight: front end of k+2 | matching of k+1 | host collects k
HBM_PEAK_GBS = 8000.0


def noisy_frames(n):
    """A short synthetic stream: one scene (seed 0), fresh sensor noise per frame."""
    import synth
    rgb0, dep0 = synth.make_frame(0, W, H)
    frames = [(rgb0, dep0)]
    for k in range(1, n):
        rng = np.random.default_rng(1000 + k)
        rgb = np.clip(rgb0.astype(np.int16) + rng.integers(-2, 3, rgb0.shape), 0, 255).astype(np.uint8)
        dn = dep0.astype(np.int32) + rng.integers(-1, 2, dep0.shape)
        dep = np.where(dep0 > 0, np.clip(dn, 300, 65535), 0).astype(np.uint16)
        frames.append((np.ascontiguousarray(rgb), np.ascontiguousarray(dep)))
    return frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--templates", type=int, default=N_TEMPLATES, help="template pyramids per object (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=["auto", "host", "device"], default="auto",
                    help="multi-GPU exchange of the match records: on the device (sharded.DeviceExchange; auto = when world > 1) or through the host")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import linemodLevelup_pybind as lm
    import sharded
    import synth

    # stdout carries exactly one line, the JSON: whatever libraries print on the way (the RCCL banner at communicator set-up) goes to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LM_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # LM_BENCH_DEVICE / LM_BENCH_BACKEND: rehearsal of the
    backend = os.environ.get("LM_BENCH_BACKEND", "nccl")                                      # world > 1 path on a 1-GPU box (gloo, all ranks on one device)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU (libamdlinemod has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.exchange == "device" and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    n_obj = max(1, world)

    det = lm.Detector(NFEAT[0], T_LEVELS, device=local_rank)
    frames = noisy_frames(N_FRAMES)
    for k, f in enumerate(frames):
        det.storeFrame(k, f)
    # quantised maps of frame 0 from the GPU front end -> planted bank (one object per rank)
    det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    det.selectFrame(0)
    det.matchResident(THRESHOLD, ["_probe"])
    quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
    classes = []
    banks = {}
    for o in range(n_obj):
        cid = "obj%02d" % o
        banks[cid] = synth.make_planted_bank(1234 + o, args.templates, quant, T_LEVELS, NFEAT)
        det.addClassPacked(cid, *banks[cid])
        classes.append(cid)
    det.setShard(rank, world)

    # Multi-GPU: the exchange of the records as device work (per-rank sort, RCCL all-gather on the exchange stream, ranking
    # merge).  Checked against the host path on one frame before anything is timed; all ranks agree on which one is used.
    ex, exchange_mode = None, "none" if world == 1 else "host"
    if args.exchange == "device" or (args.exchange == "auto" and world > 1):
        ok = 1
        try:
            ex = sharded.DeviceExchange(det, dev, force=True)
            det.selectFrame(0)
            a = sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True)
            for _ in range(2):       # a second pass if the first one outgrew the blocks (every rank sees that alike and doubles them)
                cap0 = ex.capacity
                b = sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True, exchange=ex)
                if ex.capacity == cap0:
                    break
            ok = int(a.tobytes() == b.tobytes())
        except Exception as e:   # noqa: BLE001 - any failure means the host path
            sys.stderr.write("rank %d: device exchange unavailable (%s)\n" % (rank, e))
            ok = 0
        if use_dist:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        exchange_mode = "device" if ok else "host (device exchange failed its check)"
        if not ok:
            ex = None

    host_t = {"submit": 0.0, "collect": 0.0, "gather": 0.0, "merge": 0.0}
    keys = ("frontend_ms", "coarse_ms", "local_ms", "d2h_ms", "total_ms", "coarse_candidates", "local_evals",
            "matches_pre_unique", "coarse_bytes", "local_bytes", "host_submit_ms", "host_wait_ms", "host_collect_ms", "host_merge_ms")
    acc = {k: 0.0 for k in keys}
    last = {"n": 0}

    # Pipelined stream (depth 3): the GPU prepares frame k+2 and matches frame k+1 while the host collects / sorts / gathers frame k.
    inflight_frames, redo = [], []
    xbuf = np.empty(world * 8192, lm.MATCH_DTYPE) if ex is not None else None    # the exchange writes each frame's list here

    def submit(k):
        t0 = time.perf_counter()
        det.selectFrame(k % N_FRAMES)            # device-to-device copy of a frame parked in HBM
        if ex is not None:
            ex.submit(THRESHOLD, classes)        # + sort / all-gather / merge of this frame on the exchange stream
        else:
            det.submit(THRESHOLD, classes)
        inflight_frames.append(k)
        host_t["submit"] += time.perf_counter() - t0

    def finish():
        t0 = time.perf_counter()
        k = inflight_frames.pop(0)
        if ex is not None:    # the merged, uniqued list of all ranks comes back from the device
            out = ex.collect(into=xbuf)
            t1 = t2 = t3 = time.perf_counter()
            if out is None:   # a block overflowed (same verdict on every rank): this frame is redone through the host path
                redo.append(k)
                out = np.zeros(0, lm.MATCH_DTYPE)
        elif world == 1:      # Detector.match semantics: canonical sort + unique inside the library call
            out = det.collect(sort_unique=True)
            t1 = t2 = t3 = time.perf_counter()
        else:                 # pre-unique records of this rank's shard -> all-gather -> merge on every rank
            local = det.collect(sort_unique=False, distinct=True)
            t1 = time.perf_counter()
            allrec = sharded.gather_records(local, device=dev)
            t2 = time.perf_counter()
            out = lm.merge_matches(allrec)
            t3 = time.perf_counter()
        host_t["collect"] += t1 - t0; host_t["gather"] += t2 - t1; host_t["merge"] += t3 - t2
        tm = det.lastTimings()
        for q in keys:
            acc[q] += tm[q]
        last["n"] = len(out)

    def run(nsteps):
        inflight = 0
        for k in range(nsteps):
            submit(k)
            inflight += 1
            if inflight == PIPELINE_DEPTH:
                finish()
                inflight -= 1
        while inflight:
            finish()
            inflight -= 1
        while redo:           # nothing in flight here
            k = redo.pop(0)
            ex.grow_if_needed()
            det.selectFrame(k % N_FRAMES)
            last["n"] = len(sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The roofline leg: frames one at a time, so that every kernel runs alone (in the pipelined region below the coarse pass of
    # frame k+1 and the duplicate removal of frame k-1 share the GPU with the refinement of frame k, which stretches each
    # kernel's own duration while shortening the frame).  HIP events on the kernels' stream, recorded by the library.
    excl = {"coarse_ms": 0.0, "local_ms": 0.0, "coarse_bytes": 0.0, "local_bytes": 0.0}
    EXCL = 20
    for k in range(3 + EXCL):
        det.selectFrame(k % N_FRAMES)
        det.matchResident(THRESHOLD, classes, sort_unique=False, distinct=True)
        if k >= 3:
            tm = det.lastTimings()
            for q in excl:
                excl[q] += tm[q] / EXCL

    run(args.warmup)
    fence()
    for q in host_t:
        host_t[q] = 0.0
    for q in acc:
        acc[q] = 0.0
    t0 = time.perf_counter()
    run(args.steps)                              # exactly K submits and K collects inside the timed region
    fence()
    dt = time.perf_counter() - t0
    n_final = last["n"]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    K = max(1, args.steps)
    mean = {k: acc[k] / K for k in keys}
    total_templates = args.templates * n_obj
    value = total_templates * (W * H / 1e6) * K / dt

    if rank == 0:
        # dominant kernel of this rank, timed alone (see the roofline leg above)
        if excl["local_ms"] >= excl["coarse_ms"]:
            kname, kms, kbytes, kpipe = "k_local", excl["local_ms"], excl["local_bytes"], mean["local_ms"]
        else:
            kname, kms, kbytes, kpipe = "k_coarse", excl["coarse_ms"], excl["coarse_bytes"], mean["coarse_ms"]
        achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        gbps = lambda b, ms: (b / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        out = {
            "metric": "templates*Mpixels matched/sec on 640x480 RGB-D",
            "value": value, "unit": "templates*Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[1]: 1 object x %d templates, 640x480 RGB-D, Detector(150,[4,8]), threshold 75, planted synthetic bank" % args.templates)
                                   if world == 1 else
                                   ("configs[1] scaled weakly (= configs[3] at 8 GPUs): %d objects x %d templates, bank sharded one object's worth per GPU, "
                                    "640x480 RGB-D, Detector(150,[4,8]), threshold 75, planted synthetic banks" % (n_obj, args.templates)),
                       "templates_total": total_templates, "objects": n_obj, "frames_in_stream": N_FRAMES,
                       "features_per_template": [2 * NFEAT[0], 2 * NFEAT[1]], "parallelism": "bank-shard x%d + all-gather" % world, "exchange": exchange_mode, "exchange_capacity": (ex.capacity if ex is not None else None),
                       "pipeline_depth": PIPELINE_DEPTH,
                       "coarse_candidates_per_step": mean["coarse_candidates"], "matches_pre_unique_per_step": mean["matches_pre_unique"],
                       "matches_final_last_step": n_final, "templates_per_sec": total_templates * K / dt},
            "stages_ms": {k: mean[k] for k in ("frontend_ms", "coarse_ms", "local_ms", "d2h_ms", "total_ms")},
            "host_wall_ms": dict({q: host_t[q] / K * 1e3 for q in host_t},
                                 **{q: mean[q] for q in ("host_submit_ms", "host_wait_ms", "host_collect_ms", "host_merge_ms")}),
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": kbytes, "kernel_ms": kms,
                         "duration_source": "HIP events around the kernel on its stream, %d frames submitted one at a time inside bench.py (the kernel alone on the GPU)" % EXCL,
                         "in_pipelined_region": {"kernel_ms": kpipe, "GBps": gbps(kbytes, kpipe),
                                                 "note": "same launches in the timed region, sharing the GPU with the next frame's coarse pass and front end; "
                                                         "frame-level: (coarse + local algorithmic bytes) / ms_per_step = %.0f GB/s"
                                                         % gbps(mean["coarse_bytes"] + mean["local_bytes"], dt / K * 1e3)},
                         "other": {"k_coarse_GBps": gbps(excl["coarse_bytes"], excl["coarse_ms"]), "k_local_GBps": gbps(excl["local_bytes"], excl["local_ms"])}},
        }
        if world == 1:
            out["extras"] = {"synchronous_call": sync_latency(det, classes, args.templates),
                             "pcie_inclusive": pcie_inclusive(det, frames, classes, args.templates),
                             "icp": icp_bench(local_rank),
                             "pipeline": pipeline_bench(det, frames, banks[classes[0]], classes)}
        traffic = os.path.join(ROOT, "profiles", "roofline_traffic.json")   # PMC pass of the same command (FETCH_SIZE x2 + WRITE_SIZE)
        if os.path.exists(traffic):
            try:
                tj = json.load(open(traffic))
                if tj.get("kernel") == kname:
                    out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = tj.get("source")
            except (OSError, ValueError):
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, banks[classes[0]], args.templates)
            out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def sync_latency(det, classes, n_templates, steps=20):
    """One frame at a time (submit + collect back to back, frame resident in HBM): the latency of a
    synchronous Detector.match call without the host<->device frame copy."""
    for k in range(3):
        det.selectFrame(k % N_FRAMES)
        det.matchResident(THRESHOLD, classes)
    t0 = time.perf_counter()
    for k in range(steps):
        det.selectFrame(k % N_FRAMES)
        det.matchResident(THRESHOLD, classes)
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_frame": dt * 1e3, "value": n_templates * (W * H / 1e6) / dt, "unit": "templates*Mpx/s"}


def pcie_inclusive(det, frames, classes, n_templates, steps=20):
    """Detector.match as the drop-in boundary hands it over: host numpy frames in, Match records out
    (H2D of the frame through pinned staging included).  Never the headline `value`."""
    for k in range(3):
        det.matchArray(list(frames[k % len(frames)]), THRESHOLD, classes)
    t0 = time.perf_counter()
    for k in range(steps):
        det.matchArray(list(frames[k % len(frames)]), THRESHOLD, classes)
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_frame": dt * 1e3, "value": n_templates * (W * H / 1e6) / dt, "unit": "templates*Mpx/s"}


def icp_bench(device, hypotheses=16, reps=5):
    """BASELINE configs[2]: poseRefine on the top-16 hypotheses of a frame, one ICP launch (one workgroup
    per hypothesis, <=30 point-to-plane iterations each).  Reports ICP iterations/sec."""
    import linemodLevelup_pybind as lm
    import synth
    K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
    rng = np.random.default_rng(7)
    scene_model = synth.synth_model_depth(100)
    scene = np.where(scene_model > 0, scene_model + 4, 0).astype(np.uint16)
    scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
    mds, xy = [], []
    for h in range(hypotheses):
        md = synth.synth_model_depth(100 + (h % 4))
        ys, xs = np.nonzero(md)
        mds.append(md)
        xy.append((int(xs.min()) + int(rng.integers(-2, 3)), int(ys.min()) + int(rng.integers(-2, 3))))
    Ks = np.tile(K.reshape(1, 9), (hypotheses, 1))
    Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (hypotheses, 1))
    ts = np.tile(np.array([[0, 0, 1000]], np.float32), (hypotheses, 1))
    def summary(res, dev_ms, wall):
        iters = sum(r["iterations"] for r in res if r["residual"] >= 0)
        return {"hypotheses": hypotheses, "iterations_total": iters, "device_ms": dev_ms, "wall_ms": wall * 1e3,
                "icp_iters_per_sec_device": iters / (dev_ms * 1e-3) if dev_ms > 0 else 0.0,
                "icp_iters_per_sec_wall": iters / wall, "points_source_mean": float(np.mean([r["n_source"] for r in res])),
                "points_target_mean": float(np.mean([r["n_target"] for r in res])),
                "mean_fitness": float(np.mean([r["residual"] for r in res]))}
    # (1) depth images resident in HBM (scene uploaded once per frame, model renderings in slots): the timed
    #     region is lm_icp_run = cloud preparation + normals + all ICP iterations on the device + result read-back
    ctx = lm.IcpContext(device=device, scene_from_scene=True)
    ctx.set_scene(scene, K)
    ctx.set_models(mds)
    ctx.run(Ks, Rs, ts, xy)                                   # warm-up (allocations)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        res, dev_ms = ctx.run(Ks, Rs, ts, xy)
        wall = time.perf_counter() - t0
        cur = summary(res, dev_ms, wall)
        if best is None or cur["wall_ms"] < best["wall_ms"]:
            best = cur
    ctx.close()
    # (2) the reference-shaped call on host numpy images (PCIe upload of 1 + 16 depth images inside)
    pcie = None
    for _ in range(reps):
        t0 = time.perf_counter()
        res, dev_ms = lm.pose_refine_batch(scene, K, mds, Ks, Rs, ts, xy, device=device, scene_from_scene=True)
        wall = time.perf_counter() - t0
        cur = summary(res, dev_ms, wall)
        if pcie is None or cur["wall_ms"] < pcie["wall_ms"]:
            pcie = cur
    best["pcie_inclusive"] = {"wall_ms": pcie["wall_ms"], "icp_iters_per_sec_wall": pcie["icp_iters_per_sec_wall"]}
    return best


def pipeline_bench(det, frames, bank, classes, top_k=16, steps=20):
    """BASELINE configs[2] end to end: match (2k templates) -> boxes -> NMS -> top-16 -> poseRefine on every kept match,
    one stream of device work per frame (lm_pipeline_run), everything resident in HBM.  The depth rendering of a
    template view (what the reference driver gets from its OpenGL renderer) is synthetic: the scene depth under the
    template's best match on frame 0, moved to the image centre, pushed back 3 mm and shifted 2 px, so that ICP has
    real work to do."""
    import linemodLevelup_pybind as lm
    K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
    rgb, dep = frames[0]
    feat, offs, wh = bank
    E = 2 * len(T_LEVELS)
    n = (len(offs) - 1) // E
    m = det.matchArray([rgb, dep], THRESHOLD, classes)
    best = {}
    for r in m:                                                  # canonical order: the first entry of a template is its best
        best.setdefault(int(r["template_id"]), (int(r["x"]), int(r["y"])))
    pipe = lm.Pipeline(det, W, H, scene_from_scene=True)
    R = np.eye(3, dtype=np.float32)
    chunk = 100
    for t0 in range(0, n, chunk):
        rens, Ks, Rs, ts = [], [], [], []
        for t in range(t0, min(n, t0 + chunk)):
            w, h = int(wh[t * E][0]), int(wh[t * E][1])
            x, y = best.get(t, (W // 2 - w // 2, H // 2 - h // 2))
            # the renderer puts the object at the image centre (the reference reads its depth there, LL.cpp:62)
            ren = np.zeros((H, W), np.uint16)
            patch = np.roll(dep[y:y + h, x:x + w], 2, axis=1)
            oy, ox = H // 2 - h // 2, W // 2 - w // 2
            ren[oy:oy + h, ox:ox + w] = np.where(patch > 0, patch + 3, 0)
            if ren[H // 2, W // 2] == 0:
                ren[H // 2, W // 2] = int(np.median(patch[patch > 0])) + 3 if (patch > 0).any() else 1000
            rens.append(ren); Ks.append(K); Rs.append(R); ts.append(np.array([0, 0, 1000], np.float32))
        pipe.set_views(classes[0], rens, Ks, Rs, ts, first_template=t0)
    det.setFrame([rgb, dep])
    for _ in range(3):
        res, tm = pipe.run(THRESHOLD, classes, K, top_k=top_k, nms_iou=0.5)
    acc = {"match_ms": 0.0, "nms_ms": 0.0, "icp_ms": 0.0, "total_ms": 0.0, "icp_iterations": 0}
    t0 = time.perf_counter()
    for _ in range(steps):
        res, tm = pipe.run(THRESHOLD, classes, K, top_k=top_k, nms_iou=0.5)
        for q in acc:
            acc[q] += tm[q]
    wall = (time.perf_counter() - t0) / steps
    pipe.close()
    out = {q: acc[q] / steps for q in acc}
    out.update({"wall_ms_per_frame": wall * 1e3, "frames_per_sec": 1.0 / wall, "top_k": top_k, "detections": len(res),
                "refined": int(sum(1 for r in res if r["status"] == 0)),
                "mean_fitness": float(np.mean([r["residual"] for r in res if r["status"] == 0])) if res else 0.0,
                "icp_iters_per_sec_device": (acc["icp_iterations"] / steps) / (out["icp_ms"] * 1e-3) if out["icp_ms"] > 0 else 0.0,
                "templates": n})
    return out


def cpu_baseline(frames, bank, n_templates):
    """The oracle's C (SSE2/SSSE3) port of the matching step, single thread like the reference,
    on the host cores of this box: spread/response/linearise + coarse + local for the same bank on
    the first two frames of the stream.  Quantisation (numpy in the oracle) is NOT timed, which can
    only flatter the CPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    od = lo.OracleDetector(NFEAT[0], T_LEVELS)
    feat, offs, wh = bank
    pb = lo.PackedBank(n_templates, 2, feat, offs, wh)
    times, cands = [], 0
    use = frames[:2]
    for rgb, dep in use:
        pyr = od.quantize_pyramid(rgb, dep)
        t0 = time.perf_counter()
        lms = [[lo.build_linear_memories(p[0], T_LEVELS[l]), lo.build_linear_memories(p[1], T_LEVELS[l])] for l, p in enumerate(pyr)]
        sizes = [(p[0].shape[1], p[0].shape[0]) for p in pyr]
        m, st = lo.match_bank_c(pb, lms, sizes, T_LEVELS, THRESHOLD, 1)
        times.append(time.perf_counter() - t0)
        cands += st["coarse_candidates"]
    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    lo.match_bank_c(pb, lms, sizes, T_LEVELS, THRESHOLD, ncores)
    t_mt = time.perf_counter() - t0
    sec = float(np.median(times))
    cpu = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n_templates * (W * H / 1e6) / sec, "unit": "templates*Mpx/s", "cores": 1, "kind": "port",
            "sample": "%d frames x %d templates (same bank/frames as the GPU run), SSE C port of LL.cpp:1026-1941, "
                      "linear memories + coarse + local timed, numpy quantisation excluded; median %.3f s/frame, %.1f coarse candidates/template"
                      % (len(use), n_templates, sec, cands / len(use) / n_templates),
            "host_cpu": cpu, "host_cores": ncores,
            "all_cores_variant": {"threads": ncores, "value": n_templates * (W * H / 1e6) / t_mt,
                                  "note": "templates split across pthreads, match loops only (not what the reference does)"}}


if __name__ == "__main__":
    main()
