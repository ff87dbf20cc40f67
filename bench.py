#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X linemodLevelup hot path (BASELINE.json metric:
templates·Mpixels matched/sec on 640x480 RGB-D).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

A *step* is one `Detector.match` of the reference driver loop (linemod_and_levelup_test.py:314-327,
linemod_ros/detect.py:83-138) on a NEW host frame: the frame is handed over in host memory
(lm_detector_submit_frame: pinned staging ring, H2D on a copy stream), then front end (quantise,
spread, response, bit planes) + coarse similarity over all templates + 16x16 refinement of every
candidate + duplicate removal + download, canonical sort and unique of the match records (+, for N>1,
the all-gather of the per-rank records over RCCL and the merge on the device).  Sixteen frames are in
flight and up to eight consecutive frames share their kernel launches, so the upload of later frames
overlaps the matching of earlier ones — SURVEY §8(d): "t_frame = one match call with the bank resident,
frame H2D included"; the per-call figures (frame resident / from host memory) are under
`extras.synchronous_call` / `extras.pcie_inclusive`.  No frame is replayed from HBM: the host holds a
pool of distinct noisy frames and every step stamps its number into the frame it submits.

Workload (N=1): BASELINE configs[1] — 1 object x 2000 template pyramids, Detector(150,[4,8])
(150+150 features at level 0, 75+75 at level 1), threshold 75, planted synthetic bank (tests/
synth.py).  N>1, --scaling weak (default): configs[3] shape — N objects x 2000 templates, one object
per rank; --scaling strong: a fixed bank of 8 objects x 2000 templates (configs[3], 16k) split over
the N ranks (whole objects per rank when N divides 8, contiguous template ranges otherwise).  Each
rank holds only the templates it searches; the match records are all-gathered.

Launch: N=1 runs in this process.  N>1: under torchrun (RANK / WORLD_SIZE set) this process is one
rank; started plainly (`python bench.py --gpus N`) it re-executes itself under
`python -m torch.distributed.run --standalone --nproc-per-node N`, one rank per GPU (on a box with
fewer GPUs than ranks: LM_BENCH_BACKEND=gloo LM_BENCH_DEVICE=0 rehearses the path on one device).

Before anything is timed (N=1) the GPU's results must equal the CPU oracle's — the synchronous call on two
frames AND the streamed path with shared launches on four steps — or the bench exits without a number.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel (k_local_bits): `achieved` /
`frac` = ALGORITHMIC response bytes per launch (sum nfeat*256 per 16x16 evaluation) / the
kernel's mean duration from HIP events on its stream, against the HBM peak — the convention BASELINE.json
asks for, labelled as NOT a physical fraction (the bit-plane kernels load 16 bytes where the reference
reads 256, from L1 / L2, so it exceeds 1); `binding` = the ceiling that physically binds the kernel and the
fraction of it the launch reaches (<= 1), `stages` = the same per stage of a frame (front end, coarse pass,
refinement), `traffic` = HBM bytes — all from rocprofv3 --pmc passes this script runs itself over its own
roofline leg.  `cpu_baseline` times the REFERENCE's own `Detector::match` lines (oracle/_ref, its flags,
one thread like the reference) and the oracle's SSE port on the host (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "6dpose_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))           # synth.py: the synthetic frames and banks of the workload

W, H = 640, 480
T_LEVELS = [4, 8]
NFEAT = (150, 75)
N_TEMPLATES = 2000
STRONG_OBJECTS = 8            # --scaling strong: configs[3], 8 objects x 2000 templates = 16k
THRESHOLD = 75.0
N_FRAMES = 16                 # distinct host frames in the pool the stream cycles through
N_PARKED = 4                  # frames parked in HBM for the resident legs (roofline, extras)
PIPELINE_DEPTH = int(os.environ.get("LM_BENCH_DEPTH", "0"))   # frames in flight (<= lm_detector_max_in_flight() = 16): upload, front end, coarse pass, refinement, duplicate removal of
                                                              # neighbouring BATCHES of frames side by side while the host collects the oldest frame.  Default (0): three batches
                                                              # (lm_detector_get_batch() = 4 frames share their kernel launches: 12 frames); the timed region starts and ends
                                                              # with an empty pipeline (profiles/host_profile.py: 20 steps 0.201 ms/frame at 8 in flight, 0.187 at 12)
LOAD_CYCLES_L2, LOAD_CYCLES_L1 = 31.5, 23.5   # CU cycles per 16-byte wave load, lines from L2 / all from L1 (profiles/r03_tcp_rotation_microbench.txt, patterns 0 and 7)
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0         # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
LDS_PEAK_GBS = 150000.0       # ibid. "LDS": ~150 TB/s aggregate for ds_read_b64/b128


def noisy_frames(n):
    """A synthetic stream: one scene (seed 0), fresh sensor noise per frame."""
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


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (one per GPU) and pass the JSON line through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def dry_ranks(args):
    """`--gpus N` rehearsed on one GPU: N rank processes under torch.distributed.run, gloo instead of RCCL (N
    processes cannot share one device through RCCL), every rank on device 0.  Both scalings; each run checks the device exchange
    against the host path AND the merged list against the unsharded match.  The times only prove the path (N processes time-slice
    one GPU)."""
    n = args.dry_ranks
    out = {"dry_ranks": n, "backend": "gloo", "device": 0}
    rc_all = 0
    for scaling in ("weak", "strong"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__), "--gpus", str(n), "--scaling", scaling, "--steps", str(min(args.steps, 8)),
               "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--check-unsharded", "--templates", str(args.templates)]
        env = dict(os.environ, LM_BENCH_BACKEND="gloo", LM_BENCH_DEVICE="0", OMP_NUM_THREADS="2")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        rc_all |= r.returncode
        if r.returncode != 0 or not line:
            out[scaling] = {"rc": r.returncode, "stderr_tail": r.stderr[-1500:]}
            continue
        d = json.loads(line[-1])
        out[scaling] = {"rc": 0, "ranks_observed": d["config"]["ranks_observed"], "templates_per_rank": d["config"].get("templates_per_rank"),
                        "templates_total": d["config"]["templates_total"], "exchange": d["config"]["exchange"],
                        "exchange_capacity": d["config"]["exchange_capacity"], "equals_unsharded": d["config"].get("unsharded_check"),
                        "parallelism": d["config"]["parallelism"], "ms_per_step_time_sliced": d["ms_per_step"], "value": d["value"]}
    print(json.dumps(out))
    return 1 if rc_all else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=9, help="timed regions of exactly --steps steps each; ms_per_step is their median")
    ap.add_argument("--templates", type=int, default=N_TEMPLATES, help="template pyramids per object")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=os.environ.get("LM_BENCH_SCALING", "weak"),
                    help="N>1: weak = one object x --templates per GPU (configs[1] scaled up); strong = a fixed bank of 8 objects x "
                         "--templates (configs[3]) split over the GPUs.  N=1 --scaling strong runs that 16k bank on one GPU")
    ap.add_argument("--batch", type=int, default=0, help="frames per kernel launch in stream mode (lm_detector_set_batch, 1..8; 0 = the library's default, 8)")
    ap.add_argument("--batch-queue", type=int, default=2, help="launched batches kept queued on the GPU before frames wait for a full batch (lm_detector_set_batch_queue)")
    ap.add_argument("--roofline-only", action="store_true", help="set-up + the roofline leg only (the command profiled under rocprofv3: every k_local / "
                                                                 "k_coarse launch of the run then is one of the measured launches, bar the set-up probe)")
    ap.add_argument("--check-unsharded", action="store_true", help="N > 1: rank 0 also matches frame 0 with the WHOLE bank on one detector and the sharded, "
                                                                   "gathered list has to equal it (abort otherwise)")
    ap.add_argument("--dry-ranks", type=int, default=0, help="rehearsal of `--gpus N` on ONE GPU: N processes over gloo (all on device 0), both scalings, "
                                                             "each checked against the unsharded match; prints one JSON line with both bench lines")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc passes of the roofline leg (quote the committed numbers instead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the GPU-vs-oracle comparison of frames 0 and 1 before timing (N=1)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extras legs (resident replay, ICP, pipeline, thresholds, 16k bank)")
    ap.add_argument("--exchange", choices=["auto", "host", "device"], default="auto",
                    help="multi-GPU exchange of the match records: on the device (sharded.DeviceExchange; auto = when world > 1) or through the host")
    args = ap.parse_args()
    if args.dry_ranks > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(dry_ranks(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

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
    if args.gpus != world and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s): reporting n_gpus = %d\n" % (args.gpus, world, world))
    local_rank = int(os.environ.get("LM_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # LM_BENCH_DEVICE / LM_BENCH_BACKEND: rehearsal of the
    backend = os.environ.get("LM_BENCH_BACKEND", "nccl")                                      # world > 1 path on a 1-GPU box (gloo, all ranks on one device)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU (libamdlinemod has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.exchange == "device" and "MASTER_PORT" in os.environ)
    strong = args.scaling == "strong"
    n_obj = STRONG_OBJECTS if strong else max(1, world)
    # which templates this rank searches: whole objects when the ranks divide them, else a contiguous range of the work list
    by_class = n_obj % world == 0
    my_objs = list(range(rank * n_obj // world, (rank + 1) * n_obj // world)) if by_class else list(range(n_obj))

    # The detector first, the process group after it: the HIP runtime hands a new stream the least used hardware queue of its
    # priority pool, and the detector's streams should not have to share queues with the ones torch / RCCL create (measured, world 1
    # over RCCL on one box: 0.26-0.28 ms per step with the process group first, 0.19 with the detector first)
    # and before the detector, the CPUs: this process (its helper threads, its pinned staging buffers) onto the NUMA node of its GPU —
    # what `numactl --cpunodebind` does for a deployment; LM_NO_BIND=1 leaves the placement to the scheduler
    all_cpus = os.sched_getaffinity(0)
    host_cpus = None if os.environ.get("LM_NO_BIND") else lm.bind_near_device(local_rank)
    det = lm.Detector(NFEAT[0], T_LEVELS, device=local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    if args.batch > 0:
        det.setBatch(args.batch)
    BATCH = det.getBatch()
    global PIPELINE_DEPTH
    if PIPELINE_DEPTH <= 0:
        PIPELINE_DEPTH = min(lm.load_library().lm_detector_max_in_flight(), BATCH * 3)
    frames = noisy_frames(N_FRAMES)
    for k in range(N_PARKED):
        det.storeFrame(k, frames[k])
    # quantised maps of frame 0 from the GPU front end -> planted banks (only the objects this rank searches are built and uploaded)
    det.addClassPacked("_probe", np.zeros((0, 3), np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    det.selectFrame(0)
    det.matchResident(THRESHOLD, ["_probe"])
    quant = [(det.readStage(l, 0).reshape(H >> l, W >> l), det.readStage(l, 1).reshape(H >> l, W >> l)) for l in range(2)]
    classes = ["obj%02d" % o for o in range(n_obj)]      # the class_ids of every match call: the same list on every rank (class positions are global)
    banks = {}
    for o in my_objs:
        banks[classes[o]] = synth.make_planted_bank(1234 + o, args.templates, quant, T_LEVELS, NFEAT)
        det.addClassPacked(classes[o], *banks[classes[o]])
    if not by_class:
        det.setShard(rank, world)
    my_templates = args.templates * len(my_objs) if by_class else None

    # Multi-GPU: the exchange of the records as device work (per-rank sort, RCCL all-gather on the exchange stream, ranking
    # merge).  Checked against the host path on one frame before anything is timed; all ranks agree on which one is used.
    ex, exchange_mode = None, "none" if world == 1 else "host"
    if args.exchange == "device" or (args.exchange == "auto" and world > 1):
        ok = 1
        try:
            ex = sharded.DeviceExchange(det, dev, force=True, shard=not by_class)
            det.selectFrame(0)
            a = sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True, shard=not by_class)
            for _ in range(4):       # another pass if the last one outgrew the blocks (every rank sees that alike and doubles them)
                cap0 = ex.capacity
                b = sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True, exchange=ex, shard=not by_class)
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

    # N > 1: what every rank searches, and (--check-unsharded) the sharded result of frame 0 against ONE detector holding the whole
    # bank (rank 0 builds it): the gathered, merged list must be the unsharded Detector.match list, entry by entry.
    templates_per_rank, unsharded = None, None
    if use_dist and world > 1:
        mine = args.templates * len(my_objs) if by_class else (args.templates * n_obj * (rank + 1) // world - args.templates * n_obj * rank // world)
        templates_per_rank = [None] * world
        dist.all_gather_object(templates_per_rank, int(mine))
        if args.check_unsharded:
            det.selectFrame(0)
            got = sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True, exchange=ex, shard=not by_class)
            if got is None or (ex is not None and len(got) == 0):
                got = sharded.match_sharded(det, None, THRESHOLD, classes, device=dev, resident=True, shard=not by_class)
            okf = 1
            if rank == 0:
                full = lm.Detector(NFEAT[0], T_LEVELS, device=local_rank)
                for o in range(n_obj):
                    full.addClassPacked(classes[o], *(banks[classes[o]] if classes[o] in banks else synth.make_planted_bank(1234 + o, args.templates, quant, T_LEVELS, NFEAT)))
                want = full.matchArray(list(frames[0]), THRESHOLD, classes)
                okf = int(got.tobytes() == want.tobytes())
                unsharded = {"equal": bool(okf), "matches": int(len(want)), "sharded_matches": int(len(got)), "templates": args.templates * n_obj}
                del full
            flag = torch.tensor([okf], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if rank == 0:
                    sys.stderr.write("bench.py: the sharded match differs from the unsharded one: %s\n" % json.dumps(unsharded))
                sys.exit(4)

    # Parity gate ("parity gate before any timing counts"): the GPU's Detector.match of pool frames 0 and 1
    # against the CPU oracle (quantisation in numpy, match loops = the SSE C port pinned to the reference's lines) - bit-exact
    # (x, y, similarity, template_id) in the canonical order, or the bench aborts without a number.
    parity = None
    if world == 1 and not strong and not args.no_parity_gate:
        parity = parity_gate(det, frames, banks[classes[0]], classes, args.templates)

    host_t = {"submit": 0.0, "collect": 0.0, "gather": 0.0, "merge": 0.0}
    keys = ("frontend_ms", "coarse_ms", "local_ms", "d2h_ms", "h2d_ms", "total_ms", "coarse_candidates", "local_evals",
            "matches_pre_unique", "coarse_bytes", "local_bytes", "host_submit_ms", "host_wait_ms", "host_collect_ms", "host_merge_ms", "batch_frames")
    acc = {k: 0.0 for k in keys}
    last = {"n": 0}
    tlog = {"buf": (lm.Timings * (max(1, args.steps) + 64))(), "n": 0}

    # Pipelined stream (depth 3): the GPU uploads / prepares frame k+2 and matches frame k+1 while the host collects / sorts / gathers frame k.
    inflight_frames, redo = [], []
    xbuf = {"a": None}

    def host_frame(k):
        """The frame of step k, in ordinary host memory; its number is stamped into a few pixels, so no two steps submit the same bytes."""
        rgb, dep = frames[k % N_FRAMES]
        rgb[k % H, :8, 0] = k & 0xFF
        return rgb, dep

    def submit(k, resident):
        t0 = time.perf_counter()
        frame = None
        if resident:
            det.selectFrame(k % N_PARKED)        # device-to-device copy of a frame parked in HBM (extras.resident_replay only)
        else:
            frame = host_frame(k)                # a host frame: staged + uploaded by the library (lm_detector_submit_frame)
        if ex is not None:
            ex.submit(THRESHOLD, classes, frame=frame)   # + sort / all-gather / merge of this frame on the exchange stream
        elif frame is None:
            det.submit(THRESHOLD, classes)
        else:
            det.submitFrame(frame, THRESHOLD, classes)
        inflight_frames.append((k, resident))
        host_t["submit"] += time.perf_counter() - t0

    def finish():
        t0 = time.perf_counter()
        k, resident = inflight_frames.pop(0)
        if ex is not None:    # the merged, uniqued list of all ranks comes back from the device
            if xbuf["a"] is None or len(xbuf["a"]) < world * ex.capacity:
                xbuf["a"] = np.empty(world * ex.capacity, lm.MATCH_DTYPE)
            out = ex.collect(into=xbuf["a"])
            t1 = t2 = t3 = time.perf_counter()
            if out is None:   # a block overflowed (same verdict on every rank): this frame is redone through the host path
                redo.append((k, resident))
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
        if tlog["n"] < len(tlog["buf"]):                        # the frame's lm_timings, read out after the timed region (one ctypes call here)
            det.lastTimingsInto(tlog["buf"][tlog["n"]])
            tlog["n"] += 1
        last["n"] = len(out)

    # ... and the gate on the streamed path itself (world 1: the timed loop's submit / collect with the same number of frames in flight)
    if parity is not None:
        parity["stream"] = stream_gate(det, host_frame, classes, banks[classes[0]], args.templates, PIPELINE_DEPTH, args.batch_queue)

    def run(nsteps, resident=False, first=0):
        inflight = 0
        for k in range(first, first + nsteps):
            submit(k, resident)
            inflight += 1
            if inflight == PIPELINE_DEPTH:
                finish()
                inflight -= 1
        while inflight:
            finish()
            inflight -= 1
        while redo:           # nothing in flight here
            k, res = redo.pop(0)
            ex.grow_if_needed()
            if res:
                det.selectFrame(k % N_PARKED)
            last["n"] = len(sharded.match_sharded(det, None if res else list(host_frame(k)), THRESHOLD, classes, device=dev, resident=res,
                                                  shard=not by_class))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    primed = {"live": False, "resident": False}
    rank_times = {"dt": None}

    def timed(nsteps, warmup, resident=False):
        # one-time set-up, before the W warm-up steps: every slot of the result ring (lm_detector_max_in_flight() = 16) allocates its
        # pinned staging buffer and device buffers the first time a frame lands in it - with W < 16 that would happen inside the
        # timed region (~0.5 ms per slot at the driver's --steps 20 --warmup 5)
        key = "resident" if resident else "live"
        if not primed[key]:
            run(lm.load_library().lm_detector_max_in_flight(), resident)
            primed[key] = True
        run(warmup, resident)
        fence()
        for q in host_t:
            host_t[q] = 0.0
        tlog["n"] = 0
        t0 = time.perf_counter()
        run(nsteps, resident, first=warmup)      # exactly K submits and K collects inside the timed region
        fence()
        dt = time.perf_counter() - t0
        for q in acc:
            acc[q] = 0.0
        for i in range(tlog["n"]):
            for q in keys:
                acc[q] += getattr(tlog["buf"][i], q)
        if world > 1:
            each = [None] * world                          # every rank's own time of the region: the line reports min / max besides the MAX the contract asks for,
            dist.all_gather_object(each, float(dt))        # so that the first run on a real node shows at once whether one rank (one GPU, one NUMA node) lags
            rank_times["dt"] = each
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # The roofline leg: ONE batch of frames at a time (submit BATCH host frames -> one front end / k_coarse / k_local / k_dedupe launch
    # for all of them -> collect them), so that every kernel runs alone on the GPU: in the pipelined region below the coarse pass
    # of batch k+1 and the duplicate removal of batch k-1 share the GPU with the refinement of batch k, which stretches each
    # kernel's own duration while shortening the frame.  These are the same launches (same grid, same frames per launch) as
    # in the timed region.  HIP events on the kernels' streams, recorded by the library around each launch.
    excl = {"coarse_ms": 0.0, "local_ms": 0.0, "coarse_bytes": 0.0, "local_bytes": 0.0, "frontend_ms": 0.0}
    EXCL = 20
    det.setBatchQueue(0)                                  # full batches only in this leg (the stream's default launches early while the GPU's queue is short)
    for rep in range(3 + EXCL):
        for b in range(BATCH):
            det.submitFrame(frames[(rep * BATCH + b) % N_FRAMES], THRESHOLD, classes)
        for b in range(BATCH):
            det.collect(sort_unique=False, distinct=True)
            if rep >= 3:
                tm = det.lastTimings()
                assert tm["batch_frames"] == BATCH, tm
                for q in ("coarse_bytes", "local_bytes"):             # algorithmic bytes: per frame -> summed over the launch
                    excl[q] += tm[q] / EXCL
                if b == 0:                                              # the stage times are per launch (identical for the frames of a batch)
                    for q in ("coarse_ms", "local_ms", "frontend_ms"):
                        excl[q] += tm[q] / EXCL
    det.setBatchQueue(args.batch_queue)
    if args.roofline_only:
        if rank == 0:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
            print(json.dumps({"roofline_leg": excl, "frames_per_launch": BATCH, "launches": EXCL}))
            sys.stdout.flush()
        return

    K = max(1, args.steps)
    if os.environ.get("LM_BENCH_PROFILE"):            # where the calling thread's time goes (cProfile of one timed region, to stderr)
        import cProfile, io, pstats
        pr = cProfile.Profile(); pr.enable(); timed(args.steps, args.warmup); pr.disable()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); sys.stderr.write(st.getvalue())
    # THE timed region: exactly K steps (a new host frame per step, H2D included) between two fences — repeated, because at the driver's
    # flags it is ~1.7 ms long and its run-to-run spread was as large as most optimisations: `ms_per_step` is the MEDIAN of the
    # repeats (each: W warm-up steps, fence, K timed steps, fence; N > 1: the MAX over the ranks of each repeat), min / max / all in `repeats`
    reps = max(1, args.repeats)
    dts = [timed(args.steps, args.warmup) for _ in range(reps)]
    dt = sorted(dts)[len(dts) // 2]
    steady = None
    if world == 1 and not args.no_extras and args.steps < 200:     # the same loop at 200 steps: the fill and drain of the frame pipeline amortised
        steady = timed(200, args.warmup) / 200.0 * 1e3
        timed(args.steps, args.warmup)                             # (leaves the per-step accumulators of a K-step region behind, as the fields below expect)
    n_final = last["n"]
    per_rank_ms = [t / K * 1e3 for t in rank_times["dt"]] if rank_times["dt"] else None
    mean = {k: acc[k] / K for k in keys}
    host_mean = {q: host_t[q] / K * 1e3 for q in host_t}
    total_templates = args.templates * n_obj
    value = total_templates * (W * H / 1e6) * K / dt
    replay = None
    if not args.no_extras:
        dtr = timed(args.steps, args.warmup, resident=True)
        replay = {"ms_per_step": dtr / K * 1e3, "value": total_templates * (W * H / 1e6) * K / dtr, "unit": "templates*Mpx/s",
                  "note": "round-1 headline definition: %d frames parked in HBM and replayed with a device-to-device copy, no H2D in the timed region" % N_PARKED}

    if rank == 0:
        # dominant kernel of this rank, timed alone (see the roofline leg above)
        if excl["local_ms"] >= excl["coarse_ms"]:
            # (local_ms brackets the refinement launches: k_local_bits + a k_local launch for the candidates it leaves — oversized templates —,
            # or k_local alone when the bank / geometry keeps the byte planes)
            kname, kms, kbytes, kpipe = ("k_local_bits" if det.refinesOnBitPlanes() else "k_local"), excl["local_ms"], excl["local_bytes"], mean["local_ms"]
        else:
            kname, kms, kbytes, kpipe = ("k_coarse_bits" if det.getPaths()[1] == "bits" else "k_coarse"), excl["coarse_ms"], excl["coarse_bytes"], mean["coarse_ms"]
        achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        gbps = lambda b, ms: (b / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        if world == 1 and not strong:
            workload = "configs[1]: 1 object x %d templates, 640x480 RGB-D, Detector(150,[4,8]), threshold 75, planted synthetic bank" % args.templates
        elif strong:
            workload = ("configs[3]: %d objects x %d templates = %d, bank split over %d GPU(s) (%s), 640x480 RGB-D stream, Detector(150,[4,8]), "
                        "threshold 75, planted synthetic banks" % (n_obj, args.templates, total_templates, world,
                                                                   "whole objects per rank" if by_class else "contiguous template ranges"))
        else:
            workload = ("configs[1] scaled weakly (= configs[3] at 8 GPUs): %d objects x %d templates, one object per GPU, "
                        "640x480 RGB-D stream, Detector(150,[4,8]), threshold 75, planted synthetic banks" % (n_obj, args.templates))
        out = {
            "metric": "templates*Mpixels matched/sec on 640x480 RGB-D",
            "value": value, "unit": "templates*Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "repeats": {"n": reps, "ms_per_step_min": min(dts) / K * 1e3, "ms_per_step_max": max(dts) / K * 1e3, "ms_per_step_all": [t / K * 1e3 for t in dts],
                        "ms_per_step_is": "the median of n timed regions of exactly `steps` steps each"},
            "ms_per_step_steady_200": steady,
            "kernel_us_per_frame": (excl["frontend_ms"] + excl["coarse_ms"] + excl["local_ms"]) / BATCH * 1e3,
            "kernel_us_per_frame_is": "front end + coarse pass + refinement kernels of one %d-frame launch set, each timed alone on the GPU (roofline leg), per frame" % BATCH,
            "dtype": "u8", "data": "synthetic",
            "parity_checked": bool(parity and parity["ok"]), "parity": parity,
            "config": {"workload": workload, "host_cpus": host_cpus,
                       "frame_source": "host memory, a new frame per step through lm_detector_submit_frame (pinned ring + copy stream); "
                                       "H2D inside the timed region; pool of %d distinct noisy frames, step number stamped in" % N_FRAMES,
                       "templates_total": total_templates, "objects": n_obj, "templates_this_rank": my_templates,
                       "features_per_template": [2 * NFEAT[0], 2 * NFEAT[1]],
                       "parallelism": ("bank-shard x%d (%s) + all-gather" % (world, "by object" if by_class else "by template range")),
                       "ranks_observed": (dist.get_world_size() if use_dist else 1), "backend": (backend if use_dist else None),
                       "exchange": exchange_mode, "exchange_capacity": (ex.capacity if ex is not None else None),
                       "templates_per_rank": templates_per_rank, "unsharded_check": unsharded,
                       "ms_per_step_per_rank": ({"min": min(per_rank_ms), "max": max(per_rank_ms), "all": per_rank_ms} if per_rank_ms else None),
                       "exchange_host_ms_per_step_rank0": ({"gather": host_mean["gather"], "merge": host_mean["merge"]} if world > 1 else None),
                       "pipeline_depth": PIPELINE_DEPTH, "frames_per_launch_max": BATCH, "batches_kept_queued": args.batch_queue,
                       "frames_per_launch_mean_timed": mean["batch_frames"],
                       "setup_before_warmup": "16 frames through the ingest ring (each of the library's 16 result slots allocates its pinned staging "
                                              "buffer and device buffers on first use); the timed region starts and ends with an empty pipeline, "
                                              "so at K = 20 it carries one frame latency (~0.45 ms) of fill and drain",
                       "coarse_candidates_per_step": mean["coarse_candidates"], "matches_pre_unique_per_step": mean["matches_pre_unique"],
                       "matches_final_last_step": n_final, "templates_per_sec": total_templates * K / dt},
            "stages_ms": dict({k: mean[k] for k in ("h2d_ms", "frontend_ms", "coarse_ms", "local_ms", "d2h_ms", "total_ms")},
                              note="device time per LAUNCH in the pipelined region; a launch serves frames_per_launch = %d frames (h2d_ms: per frame)" % BATCH),
            "host_wall_ms": dict(host_mean, **{q: mean[q] for q in ("host_submit_ms", "host_wait_ms", "host_collect_ms", "host_merge_ms")}),
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": kbytes, "kernel_ms": kms, "frames_per_launch": BATCH,
                         "kernel_ms_covers": ("the events bracket the refinement of a batch: k_local_bits (and, for banks whose candidates can leave their planes, the k_local "
                                              "launch that takes those; none at this workload)" if kname == "k_local_bits" else "one launch of the kernel"),
                         "convention": "ALGORITHMIC bytes (SURVEY 8d: one byte per response read the reference performs) per launch / kernel time, "
                                       "against the HBM peak as BASELINE.json's metric asks.  The linear memories are cache-resident, so this is not "
                                       "physical HBM traffic (see `traffic`); the ceilings that physically bound the kernel are below",
                         "vs_cache_ceilings": {"l2_peak_GBps": L2_PEAK_GBS, "frac_of_l2": achieved / L2_PEAK_GBS,
                                               "lds_peak_GBps": LDS_PEAK_GBS, "frac_of_lds": achieved / LDS_PEAK_GBS},
                         "duration_source": "HIP events around the kernel on its stream, %d launches of %d frames each, one launch set at a time inside bench.py (the kernel alone on the GPU)" % (EXCL, BATCH),
                         "in_pipelined_region": {"kernel_ms": kpipe, "GBps": gbps(kbytes, kpipe),
                                                 "note": "same launches in the timed region, sharing the GPU with the next frame's coarse pass and front end; "
                                                         "frame-level: (coarse + local algorithmic bytes) / ms_per_step = %.0f GB/s"
                                                         % gbps(mean["coarse_bytes"] + mean["local_bytes"], dt / K * 1e3)},
                         "other": {"k_coarse_GBps": gbps(excl["coarse_bytes"], excl["coarse_ms"]), "k_local_GBps": gbps(excl["local_bytes"], excl["local_ms"]),
                                   "k_coarse_ms": excl["coarse_ms"], "k_local_ms": excl["local_ms"]}},
        }
        if replay is not None:
            out["extras"] = {"resident_replay": replay}
        if world == 1 and not strong and not args.no_extras:
            cls0 = classes[0]
            out["extras"].update({"synchronous_call": sync_latency(det, classes, args.templates),
                                  "pcie_inclusive": pcie_inclusive(det, frames, classes, args.templates),
                                  "one_candidate_per_template": sparse_threshold_run(det, frames, classes, args.templates),
                                  "icp": icp_bench(local_rank),
                                  "real_fixture": real_fixture_leg(local_rank),
                                  "pipeline": pipeline_bench(det, frames, banks[cls0], classes)})
            if not args.no_cpu_baseline:
                out["extras"]["icp"]["cpu_baseline"] = icp_cpu_baseline()
                cb = out["extras"]["icp"]["cpu_baseline"]
                out["extras"]["icp"]["speedup_vs_cpu_numpy"] = (out["extras"]["icp"]["icp_iters_per_sec_device"] / cb["icp_iters_per_sec"]) if cb["icp_iters_per_sec"] else None
            pl = out["extras"]["pipeline"]
            ic = out["extras"]["icp"]
            gbs = ic["algorithmic_bytes"] / (ic["device_ms"] * 1e-3) / 1e9 if ic["device_ms"] > 0 else 0.0
            tfs = ic["brute_force_flops"] / (ic["device_ms"] * 1e-3) / 1e12 if ic["device_ms"] > 0 else 0.0
            out["extras"]["roofline_icp"] = {
                "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "algorithmic_bytes": ic["algorithmic_bytes"], "device_ms": ic["device_ms"], "iterations": ic["iterations_total"],
                "convention": "SURVEY 8d: per ICP iteration and hypothesis N_src * 24 B + N_tgt * 48 B (every point and normal once), summed over the iterations "
                              "executed, over the device time of the whole poseRefine batch (clouds, voxel grid, kNN normals and all evaluations)",
                "vs_f64_vector_peak": {"achieved": tfs, "peak": 78.6, "unit": "TFLOP/s", "frac": tfs / 78.6,
                                       "convention": "SURVEY 8d: the flops of a brute-force nearest-neighbour search, N_src * N_tgt * 8 per iteration (the kernels "
                                                     "search a grid and execute a small fraction of them)"},
                "what_binds": "neither: the chain of DEPENDENT f64 operations of a single wave - a dependent f64 op costs a lone wave ~40 cycles on gfx950, an "
                              "LDS read 70, a permute 78 (profiles/r06_latency_microbench.txt) - through 31 evaluations of the hypotheses that never converge: "
                              "k_icp_team (a team of workgroups per hypothesis, clouds in LDS and registers; one launch, two for a batch whose clouds could use "
                              "more workgroups than the chip has) spends ~17 us per evaluation on exchange, 6x6 solve, transform + certification, search "
                              "sweep and sums (profiles/r06_icp_*)",
                "pipeline_leg": {"icp_ms": pl["icp_ms"], "iterations": pl["icp_iterations"],
                                 "achieved": (sum(i * (a * 24 + b * 48) for i, a, b in zip(pl["iterations"], pl["points_source"], pl["points_target"])) / (pl["icp_ms"] * 1e-3) / 1e9) if pl["icp_ms"] > 0 else 0.0,
                                 "unit": "GB/s"}}
        # PMC counters of the dominant kernel, measured NOW: separate rocprofv3 --pmc passes over `bench.py --roofline-only` (the
        # same launches as the roofline leg above), corrected as MI355X_MICROARCH.md (HBM) prescribes.  Without rocprofv3 (or with
        # --no-pmc) the numbers of the last committed pass (profiles/roofline_traffic.json) are quoted and labelled as such.
        paths = det.getPaths()
        k_refine = "k_local_bits" if paths[0] == "bits" else "k_local"
        k_coarse_name = "k_coarse_bits" if paths[1] == "bits" else "k_coarse"
        pm_all = None if (args.no_pmc or args.no_extras or world != 1 or strong) else pmc_live((k_refine, k_coarse_name, "k_fe_stage", "k_fe_bits", "k_pack_bits", "k_pack_top", "k_dedupe"), args)
        pm = pm_all.get(kname) if pm_all else None
        rf = out["roofline"]
        rf["frac_is"] = ("ALGORITHMIC bytes over the HBM peak (SURVEY 8d's convention): NOT a physical fraction - the bit-plane kernels load 16 bytes where the "
                         "reference reads 256, from L1 / L2 - and it exceeds 1.  What the kernel reaches of the ceilings that can bind it is `binding` (<= 1), "
                         "per stage under `stages`")
        if pm is not None:
            cz = ceilings_of(pm)
            hbm = cz["hbm_bytes_per_dispatch"]      # FETCH_SIZE / WRITE_SIZE are in KB; gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
            pairs = kbytes / 256.0                   # (candidate, feature) pairs per launch: 256 algorithmic bytes each
            lane_features = pairs / 8.0              # a wave instruction of k_local_bits serves 8 pairs (8 lanes per candidate)
            rf.update({
                "traffic": hbm,
                "traffic_source": "live: rocprofv3 --kernel-trace --pmc passes of `bench.py --roofline-only` run by this bench.py (FETCH_SIZE x2 + WRITE_SIZE, "
                                  "mean per %s dispatch of %d frames); memory-side (fabric) bytes, Infinity-Cache hits included" % (kname, BATCH),
                "binding": cz["binding"], "ceilings": cz["fractions"], "tcp_accesses_per_cu_cycle_raw": cz["tcp_accesses_per_cu_cycle_raw"],
                "tcp_miss_stall_cycles_per_cu_cycle": cz["tcp_miss_stall_cycles_per_cu_cycle"],
                "hbm_frac_physical": cz["fractions"]["hbm"],
                "kernel_us_profiled": cz["kernel_us_profiled"], "l2_hit_rate": cz["l2_hit_rate"],
                "cu_cycles_per_wave_load": cz["cu_cycles_per_wave_load"], "l1_accesses_per_wave_load": cz["l1_accesses_per_wave_load"],
                "insts_per_launch": {"salu": pm["SQ_INSTS_SALU"], "valu": pm["SQ_INSTS_VALU"], "vmem_rd": pm["SQ_INSTS_VMEM_RD"], "waves": pm["SQ_WAVES"]}})
            if kname == "k_local_bits" and lane_features > 0:
                per = pm["SQ_INSTS_VALU"] / lane_features
                rf.update({
                    "valu_per_lane_feature": per,
                    "useful_valu_frac": USEFUL_ADDER_OPS_PER_LANE_FEATURE / per,
                    "useful_valu_frac_with_window_shifts": (USEFUL_ADDER_OPS_PER_LANE_FEATURE + ALIGN_OPS_PER_LANE_FEATURE) / per,
                    "useful_valu_note": "adder operations the bit-sliced sums need per lane and feature (5.0: ISA of the 16-feature loop body, DESIGN.md 3.1) over the "
                                        "wave-level VALU instructions measured per lane and feature (SQ_INSTS_VALU / (pairs / 8)); round 3's kernel: 10 of 33",
                    "what_bounds_it": "the number of wave loads: a wave load of k_local_bits (8 candidates x 128 contiguous bytes = 1 KB) takes 20-23 CU cycles whatever "
                                      "the lanes or addresses (`cu_cycles_per_wave_load`; 16 = the 64 B per cycle of the L1 -> register path, `ceilings.l1_data`), with "
                                      "the L2 -> L1 fills behind it (`ceilings.l2`, the miss-stall share).  Three schemes that share loads between neighbouring candidates "
                                      "were built in round 5 — bit-exact, fewer wave loads, more L2 requests, none faster: profiles/r05_local_sharing/README.txt"})
        # every stage of a frame with the ceiling that binds it, from the same PMC passes (one fraction <= 1 per stage, recomputable from profiles/r04_pmc.txt)
        if pm_all:
            B_FRONT = W * H * 3 + W * H * 2 + 8 * 2 * (W * H + (W // 2) * (H // 2))      # 1.54 MB in + 6.14 MB of linear memories (8 labels x 2 modalities x 2 levels) out per VGA frame
            stages = {}
            fe_us, fe_bytes = 0.0, 0.0
            fe_parts = {}
            for k in ("k_fe_stage", "k_fe_bits", "k_pack_bits", "k_pack_top"):
                if k in pm_all:
                    cz = ceilings_of(pm_all[k], wide_loads=False)
                    n = pm_all[k]["dispatches"] / max(1, pm_all[k_refine]["dispatches"] if k_refine in pm_all else 1)     # dispatches of this kernel per batch
                    fe_us += cz["kernel_us_profiled"] * n
                    fe_bytes += cz["hbm_bytes_per_dispatch"] * n
                    fe_parts[k] = {"dispatches_per_batch": n, "us_per_dispatch": cz["kernel_us_profiled"], "binding": cz["binding"], "ceilings": cz["fractions"],
                                   "tcp_miss_stall_cycles_per_cu_cycle": cz["tcp_miss_stall_cycles_per_cu_cycle"]}
            if fe_us > 0:
                stages["frontend"] = {"kernels": fe_parts, "us_per_batch_profiled": fe_us, "ms_per_batch_events": excl["frontend_ms"],
                                      "algorithmic_bytes_per_frame": B_FRONT,
                                      "hbm_frac_algorithmic": B_FRONT * BATCH / (excl["frontend_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if excl["frontend_ms"] > 0 else None,
                                      "traffic_bytes_per_batch": fe_bytes, "traffic_over_algorithmic": fe_bytes / (B_FRONT * BATCH),
                                      "note": "the one stage that streams: a frame in, its response maps out.  With bit planes for both passes the maps leave as 2 bits per "
                                              "cell (2.8 MB per frame) instead of bytes in two layouts (11 MB), so the traffic can fall below SURVEY 8(d)'s algorithmic figure"}
            for stage, k, ms in (("coarse", k_coarse_name, excl["coarse_ms"]), ("refine", k_refine, excl["local_ms"])):
                if k in pm_all:
                    cz = ceilings_of(pm_all[k])
                    stages[stage] = {"kernel": k, "ms_per_batch_events": ms, "us_per_dispatch_profiled": cz["kernel_us_profiled"], "binding": cz["binding"],
                                     "ceilings": cz["fractions"], "tcp_accesses_per_cu_cycle_raw": cz["tcp_accesses_per_cu_cycle_raw"],
                                     "tcp_miss_stall_cycles_per_cu_cycle": cz["tcp_miss_stall_cycles_per_cu_cycle"], "l2_hit_rate": cz["l2_hit_rate"], "cu_cycles_per_wave_load": cz["cu_cycles_per_wave_load"],
                                     "l1_accesses_per_wave_load": cz["l1_accesses_per_wave_load"]}
            rf["stages"] = stages
        if pm is None:
            traffic = os.path.join(ROOT, "profiles", "roofline_traffic.json")   # last committed PMC pass (profiles/pmc_run.sh)
            if os.path.exists(traffic):
                try:
                    tj = json.load(open(traffic))
                    if tj.get("kernel") == kname:
                        rf["traffic"] = tj.get("hbm_bytes_per_launch")
                        rf["traffic_source"] = "NOT measured by this run - committed pass: " + str(tj.get("source"))
                        for q in ("binding", "ceilings", "hbm_frac_physical", "frames_per_launch"):
                            if q in tj:
                                rf.setdefault(q if q != "frames_per_launch" else "traffic_frames_per_launch", tj[q])
                except (OSError, ValueError):
                    pass
        if world == 1 and not strong and not args.no_cpu_baseline:
            os.sched_setaffinity(0, all_cpus)          # the CPU baseline (its all-cores variant) gets every core the process started with
            out["cpu_baseline"] = cpu_baseline(noisy_frames(N_FRAMES), banks[classes[0]], args.templates)
            out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
            out["speedup_vs_cpu_1thread_is"] = ("GPU frame (quantisation + matching + sort) over the CPU's MATCH LOOPS ONLY (the reference's lines from the quantised "
                                                "maps on: OpenCV, which the reference's quantisation needs, is not in this image)")
            iq = out["cpu_baseline"].get("incl_quantisation", {}).get("value")
            out["speedup_vs_cpu_1thread_incl_quantisation"] = value / iq if iq else None
        if world == 1 and not strong and not args.no_extras:
            sr = strong_reference(det, quant, frames, args.templates, steps=args.steps, depth=PIPELINE_DEPTH)
            out["extras"]["strong_scaling_reference"] = sr
            # One-GPU proxy of BASELINE's strong-scaling target (>= 6 x at 8 GPUs when sharding 16k templates): the 16k bank on this GPU against an
            # eighth of it — the headline's workload — at EQUAL step counts and frames in flight.  It bounds the 8-GPU ratio from above: the exchange
            # of the records and whatever eight host processes cost each other come on top.  Nothing here is a measured scaling curve.
            out["extras"]["strong_scaling_proxy"] = {
                "t_16k_one_gpu_ms": sr["ms_per_step"], "t_2k_share_ms": dt / K * 1e3, "steps": args.steps, "ratio": sr["ms_per_step"] / (dt / K * 1e3) if dt > 0 else None,
                "target": 6.0, "note": "T(8 x 2000 templates on one GPU) / T(2000 templates on one GPU), same stream loop, same steps; an upper bound of the 8-GPU speed-up of configs[3], "
                                       "not a measurement of it (no multi-GPU node has been available to any round)"}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def oracle_matches(od, lo, pb, rgb, dep, threshold):
    """Detector.match of the CPU oracle for one frame: numpy quantisation + the C port of the match loops + canonical merge.
    Returns (records, seconds of quantisation, seconds of linear memories + match loops)."""
    t0 = time.perf_counter()
    pyr = od.quantize_pyramid(rgb, dep)
    t1 = time.perf_counter()
    lms = [[lo.build_linear_memories(p[0], od.T_at_level[l]), lo.build_linear_memories(p[1], od.T_at_level[l])] for l, p in enumerate(pyr)]
    sizes = [(p[0].shape[1], p[0].shape[0]) for p in pyr]
    raw, st = lo.match_bank_c(pb, lms, sizes, od.T_at_level, threshold, 1)
    t2 = time.perf_counter()
    return lo.canonical_sort_unique(raw), raw, st, pyr, t1 - t0, t2 - t1


def same_records(got, want):
    return (len(got) == len(want) and np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"]) and
            np.array_equal(got["template_id"], want["tid"]) and np.array_equal(got["similarity"], want["sim"]))


def parity_gate(det, frames, bank, classes, n_templates, n_frames=2):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    od = lo.OracleDetector(NFEAT[0], T_LEVELS)
    pb = lo.PackedBank(n_templates, 2, *bank)
    out = {"ok": True, "frames": [], "oracle": "oracle/linemod_oracle.py quantisation + oracle/match_oracle.c (SSE port pinned to the reference's own "
                                               "lines, tests/test_ref_pin.py) + canonical merge; compared: x, y, similarity bits, template_id, in order"}
    for k in range(n_frames):
        rgb, dep = frames[k]
        want, _, st, _, _, _ = oracle_matches(od, lo, pb, rgb, dep, THRESHOLD)
        got = det.matchArray([rgb, dep], THRESHOLD, classes)
        tm = det.lastTimings()
        ok = same_records(got, want) and int(tm["coarse_candidates"]) == int(st["coarse_candidates"])
        out["frames"].append({"frame": k, "matches": int(len(want)), "gpu_matches": int(len(got)), "coarse_candidates": int(st["coarse_candidates"]),
                              "gpu_coarse_candidates": int(tm["coarse_candidates"]), "equal": bool(ok)})
        out["ok"] = out["ok"] and ok
    if not out["ok"]:
        sys.stderr.write("bench.py: PARITY GATE FAILED - GPU matches differ from the CPU oracle: %s\n" % json.dumps(out["frames"]))
        sys.exit(3)
    return out


def stream_gate(det, host_frame, classes, bank, n_templates, depth, batch_queue, steps=(0, 1, 6, 11), region_steps=(0, 2, 3, 12, 18, 19), region_n=20):
    """The same comparison for the path the number is measured on: host frames through lm_detector_submit_frame, `depth` in flight, the
    library's batching (several frames per kernel launch, frame -> XCD affinity) — the lists collect() returns for a few steps of a stream
    against the oracle's lists for exactly those (stamped) frames.  Two passes: full batches only (batch queue 0: every frame shares its
    launches), and the timed region's OWN launch rule (`batch_queue` as timed, a burst of `region_n` submits behind a fence: an early partial
    batch, full batches, a short last one — the 3 + 8 + 8 + 1 of the driver's 20 steps).  Exits without a number on a mismatch."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    import torch
    od = lo.OracleDetector(NFEAT[0], T_LEVELS)
    pb = lo.PackedBank(n_templates, 2, *bank)

    def burst(n, first):
        infl = 0
        for k in range(n):
            det.submitFrame(host_frame(first + k), THRESHOLD, classes)
            infl += 1
            if infl == depth:
                det.collect(sort_unique=True); infl -= 1
        while infl:
            det.collect(sort_unique=True); infl -= 1

    def one_pass(bq, want_steps, n, first):
        kept, got, infl = {}, [], 0
        det.setBatchQueue(bq)
        burst(depth + 5, first + 1000)                        # warm-up steps in a tight loop (the oracle runs of the pass before were a long pause: the library's pace estimate), ...
        torch.cuda.synchronize()                              # ... then the burst behind a fence, like the timed region
        for k in range(n):
            rgb, dep = host_frame(first + k)
            if k in want_steps:
                kept[k] = (rgb.copy(), dep.copy())            # the pool frame is stamped again by later steps
            det.submitFrame((rgb, dep), THRESHOLD, classes)
            infl += 1
            if infl == depth:
                got.append((det.collect(sort_unique=True), det.lastTimings()["batch_frames"])); infl -= 1
        while infl:
            got.append((det.collect(sort_unique=True), det.lastTimings()["batch_frames"])); infl -= 1
        res, ok_all = [], True
        for k in want_steps:
            want, _, st, _, _, _ = oracle_matches(od, lo, pb, kept[k][0], kept[k][1], THRESHOLD)
            ok = same_records(got[k][0], want)
            res.append({"step": k, "matches": int(len(want)), "gpu_matches": int(len(got[k][0])), "frames_in_its_launch": int(got[k][1]), "equal": bool(ok)})
            ok_all = ok_all and ok
        return res, ok_all, [int(g[1]) for g in got]

    out = {"ok": True, "frames_in_flight": depth}
    # full batches only: the gate must see frames that SHARE their launches, whatever the pace estimate says this early
    out["steps"], ok_a, _ = one_pass(0, steps, max(steps) + 1, 0)
    # ... and the launch shapes of the timed region itself
    reg, ok_b, shapes = one_pass(batch_queue, region_steps, region_n, 100)
    out["timed_region_rule"] = {"batch_queue": batch_queue, "submits": region_n, "steps": reg, "frames_in_the_launch_of_each_step": shapes}
    det.setBatchQueue(batch_queue)
    out["ok"] = ok_a and ok_b
    if not out["ok"]:
        sys.stderr.write("bench.py: PARITY GATE FAILED on the streamed path - collect() differs from the CPU oracle: %s\n" % json.dumps(out))
        sys.exit(3)
    return out


PMC_PASSES = (("FETCH_SIZE", "GRBM_GUI_ACTIVE"),
              ("WRITE_SIZE", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum"),
              ("SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_WAVES", "TCC_HIT_sum", "TCC_MISS_sum"))      # (MI355X_MICROARCH.md, PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass)


def pmc_live(kernels, args, timeout=150):
    """{kernel: {"dispatches": n, counter: mean per dispatch}} of the counters in PMC_PASSES for every kernel of `kernels` that ran, one rocprofv3
    run per pass (SQ, TCC and TCP counters do not all fit one pass; --kernel-trace + --pmc only, nothing else traced) over
    `python bench.py --roofline-only`.  None if rocprofv3 is not there or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    vals = {k: {} for k in kernels}
    tmp = tempfile.mkdtemp(prefix="lm_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for i, grp in enumerate(PMC_PASSES):
            outdir = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--kernel-trace", "--pmc"] + list(grp) + ["-d", outdir, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--roofline-only",
                                                                  "--no-parity-gate", "--templates", str(args.templates)] + (["--batch", str(args.batch)] if args.batch > 0 else [])
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            dbs = glob.glob(os.path.join(outdir, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            con = sqlite3.connect(dbs[0])
            cols = [c[1] for c in con.execute("pragma table_info('counters_collection')")]
            name_col = "kernel_name" if "kernel_name" in cols else "name"
            rows = con.execute("select %s, counter_name, dispatch_id, sum(value) from counters_collection group by %s, counter_name, dispatch_id" % (name_col, name_col)).fetchall()
            con.close()
            per = {k: {} for k in kernels}
            for kn, cname, _disp, val in rows:
                for k in kernels:
                    # exactly this kernel, demangled or mangled, plain or a template instantiation: "k_local(" is not "k_local_bits<5, 4>("
                    if any(t in kn for t in (k + "(", k + "<", "%d%sE" % (len(k), k), "%d%sI" % (len(k), k))):
                        per[k].setdefault(cname, []).append(val)
            for k in kernels:
                for cname, v in per[k].items():
                    vals[k][cname] = sum(v) / len(v)
                    vals[k]["dispatches"] = len(v)
        need = [c for grp in PMC_PASSES for c in grp]
        return {k: v for k, v in vals.items() if all(c in v for c in need)}
    except (OSError, subprocess.SubprocessError, sqlite3.Error):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# VALU instructions per lane and feature the bit-sliced sums need (ISA of k_local_bits' 16-feature loop body): two dwords x
# 2.5 adder operations (7 carry-save adders per 8 inputs, one more per 16, a 5-level ripple per 16: 40 v_bitop3 per dword and 16 features)
USEFUL_ADDER_OPS_PER_LANE_FEATURE = 5.0
ALIGN_OPS_PER_LANE_FEATURE = 2.0      # + the two v_alignbit that cut the lane's two window rows out of their records


def ceilings_of(pm, wide_loads=True):
    """What a kernel's launch reached of each physical ceiling, from its PMC means:
    l1_data = the vector L1 -> register path: a wave load of 16 bytes per lane (what the matching kernels issue) is 1 KB at 64 B per CU cycle = 16 cycles,
          so the fraction is 16 x wave loads over the CU cycles of the launch.  This is what binds k_local_bits: its time follows the NUMBER of wave loads
          (profiles/r05_local_sharing/README.txt; 20-23 CU cycles per wave load in every lane layout tried);
    tcp = vector-L1 (TCP) tag accesses per CU cycle, reported for reference only: round 4 read "one 64-byte access per cycle" as the ceiling, but the
          counter stands at 1.09 per cycle on the faster grid of round 5 (22.4 accesses per wave load whatever the addresses), so it is not a limit;
    valu = wave-level VALU instructions x 4 cycles over the SIMD cycles (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycle: profiles/r04_pmc.txt);
    l2 = L2 requests x 128 B (an upper bound on the bytes they move) per second over the L2 peak; hbm = memory-side bytes per second over the HBM peak."""
    kc = pm["GRBM_GUI_ACTIVE"] / 8.0
    if kc <= 0:
        return None
    us = kc / 2400.0
    hbm_bytes = 2.0 * pm["FETCH_SIZE"] * 1024.0 + pm["WRITE_SIZE"] * 1024.0
    tcp_raw = pm["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * kc)
    # l1_data assumes every vector read is a 16-byte-per-lane wave load: true of k_local_bits / k_coarse_bits (wide_loads), not of the front end and the
    # duplicate removal, whose loads are narrower - for those the ceiling is not applicable and is left out of the choice of the binding one
    c = {"l1_data": min(1.0, 16.0 * pm["SQ_INSTS_VMEM_RD"] / (256.0 * kc)) if wide_loads else None,
         "tcp": tcp_raw,
         "valu": 4.0 * pm["SQ_INSTS_VALU"] / (1024.0 * kc),
         "l2": (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"]) * 128.0 / (us * 1e-6) / 1e9 / L2_PEAK_GBS,
         "hbm": hbm_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    top = max((q for q in c if q != "tcp" and c[q] is not None), key=lambda q: c[q])
    names = {"l1_data": "vector L1 -> register path: 16-byte-per-lane wave loads at 64 B per CU cycle (16 cycles each)", "valu": "VALU issue slots (4 cycles per wave instruction)",
             "l2": "L2 bandwidth (requests x 128 B against %.1f TB/s)" % (L2_PEAK_GBS / 1e3), "hbm": "HBM bandwidth (memory-side bytes against %.0f TB/s)" % (HBM_PEAK_GBS / 1e3)}
    return {"kernel_us_profiled": us, "dispatches_profiled": pm.get("dispatches"), "fractions": c, "binding": {"ceiling": names[top], "frac": c[top]},
            "tcp_accesses_per_cu_cycle_raw": tcp_raw, "tcp_miss_stall_cycles_per_cu_cycle": pm["TCP_PENDING_STALL_CYCLES_sum"] / (256.0 * kc),
            "hbm_bytes_per_dispatch": hbm_bytes, "wave_loads": pm["SQ_INSTS_VMEM_RD"], "valu_insts": pm["SQ_INSTS_VALU"],
            "cu_cycles_per_wave_load": (256.0 * kc / pm["SQ_INSTS_VMEM_RD"]) if pm["SQ_INSTS_VMEM_RD"] else None,
            "l1_accesses_per_wave_load": (pm["TCP_TOTAL_CACHE_ACCESSES_sum"] / pm["SQ_INSTS_VMEM_RD"]) if pm["SQ_INSTS_VMEM_RD"] else None,
            "l2_hit_rate": pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"]) if (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"]) else None}


def stream_shape(det, tm):
    """What an extras leg ran on, so that its number can be set beside the headline's: kernel paths, frames per launch, device time per frame."""
    nb = max(1.0, float(tm.get("batch_frames", 1.0)))
    return {"paths": list(det.getPaths()), "frames_per_launch_mean": nb, "frames_in_flight": PIPELINE_DEPTH,
            "kernels_ms_per_frame": {q: (tm.get(q + "_ms", 0.0) / nb) for q in ("frontend", "coarse", "local")}}


def pipelined_host_stream(det, frames, classes, threshold, steps, warmup=16, depth=None):
    """Seconds per frame of the live-stream path (a host frame per step, `depth` in flight: the headline's PIPELINE_DEPTH unless given) + mean timings."""
    depth = depth or PIPELINE_DEPTH or 16
    acc, n = {}, 0
    def go(k0, cnt, record):
        nonlocal n
        infl = 0
        for k in range(k0, k0 + cnt):
            det.submitFrame(frames[k % len(frames)], threshold, classes)
            infl += 1
            if infl == depth:
                det.collect(); infl -= 1
                if record:
                    tm = det.lastTimings(); n += 1
                    for q, v in tm.items():
                        acc[q] = acc.get(q, 0.0) + v
        while infl:
            det.collect(); infl -= 1
    go(0, warmup, False)
    t0 = time.perf_counter()
    go(warmup, steps, True)
    dt = (time.perf_counter() - t0) / steps
    return dt, {q: v / max(1, n) for q, v in acc.items()}


def sparse_threshold_run(det, frames, classes, n_templates, steps=50):
    """"a second run at a threshold chosen to give ~1 candidate/template".  The threshold is found by bisection
    on the coarse candidate count of frame 0; then the same live-stream loop as the headline."""
    lo_t, hi_t = THRESHOLD, 100.0
    det.setFrame(list(frames[0]))
    for _ in range(12):
        mid = 0.5 * (lo_t + hi_t)
        det.matchResident(mid, classes, sort_unique=False, distinct=True)
        c = det.lastTimings()["coarse_candidates"]
        if c > n_templates:
            lo_t = mid
        else:
            hi_t = mid
    thr = hi_t
    dt, tm = pipelined_host_stream(det, frames, classes, thr, steps)
    return {"threshold": thr, "coarse_candidates_per_template": tm.get("coarse_candidates", 0.0) / n_templates,
            "ms_per_frame": dt * 1e3, "value": n_templates * (W * H / 1e6) / dt, "unit": "templates*Mpx/s", "steps": steps,
            "coarse_ms": tm.get("coarse_ms"), "local_ms": tm.get("local_ms"), "frontend_ms": tm.get("frontend_ms"), **stream_shape(det, tm)}


def strong_reference(det0, quant, frames, per_object, steps=20, depth=8):
    """The N=1 point of the strong-scaling curve (`--scaling strong`: 8 objects x 2000 templates on ONE GPU), so that the
    "x at 8 GPUs over 1 GPU when sharding >= 16k templates" ratio of BASELINE.json can be formed from two bench lines."""
    import linemodLevelup_pybind as lm
    import synth
    det = lm.Detector(NFEAT[0], T_LEVELS, device=0 if det0 is None else det0.device)
    classes = []
    for o in range(STRONG_OBJECTS):
        cid = "obj%02d" % o
        det.addClassPacked(cid, *synth.make_planted_bank(1234 + o, per_object, quant, T_LEVELS, NFEAT))
        classes.append(cid)
    total = per_object * STRONG_OBJECTS
    # parity on this leg's own bank and frame: two of its eight objects (first, last) against the CPU oracle, record by record
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    od = lo.OracleDetector(NFEAT[0], T_LEVELS)
    par = {"ok": True, "classes": []}
    rgb0, dep0 = frames[0]
    for o in (0, STRONG_OBJECTS - 1):
        pb = lo.PackedBank(per_object, 2, *synth.make_planted_bank(1234 + o, per_object, quant, T_LEVELS, NFEAT))
        want, _, st, _, _, _ = oracle_matches(od, lo, pb, rgb0, dep0, THRESHOLD)
        got = det.matchArray([rgb0, dep0], THRESHOLD, [classes[o]])
        ok = same_records(got, want) and int(det.lastTimings()["coarse_candidates"]) == int(st["coarse_candidates"])
        par["classes"].append({"class": classes[o], "matches": int(len(want)), "coarse_candidates": int(st["coarse_candidates"]), "equal": bool(ok)})
        par["ok"] = par["ok"] and ok
    if not par["ok"]:
        sys.stderr.write("bench.py: PARITY FAILED on the 16k bank: %s\n" % json.dumps(par))
        sys.exit(3)
    dt, tm = pipelined_host_stream(det, frames, classes, THRESHOLD, steps, warmup=16, depth=depth)
    return {"templates_total": total, "ms_per_step": dt * 1e3, "steps": steps, "value": total * (W * H / 1e6) / dt, "unit": "templates*Mpx/s", **stream_shape(det, tm),
            "workload": "configs[3] on ONE GPU: %d objects x %d templates, 640x480 stream" % (STRONG_OBJECTS, per_object), "parity": par,
            "kernel_ms_per_launch": {q: tm.get(q) for q in ("frontend_ms", "coarse_ms", "local_ms")}, "frames_per_launch": tm.get("batch_frames"),
            "coarse_ms": tm.get("coarse_ms"), "local_ms": tm.get("local_ms"), "coarse_candidates": tm.get("coarse_candidates"),
            "note": "same live-stream loop as the headline (host frame per step), `python bench.py --scaling strong` gives the same number as a bench line"}


def sync_latency(det, classes, n_templates, steps=20):
    """One frame at a time (submit + collect back to back, frame resident in HBM): the latency of a
    synchronous Detector.match call without the host<->device frame copy."""
    for k in range(3):
        det.selectFrame(k % N_PARKED)
        det.matchResident(THRESHOLD, classes)
    t0 = time.perf_counter()
    for k in range(steps):
        det.selectFrame(k % N_PARKED)
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
    """BASELINE configs[2]: poseRefine on the top-16 hypotheses of a frame (cloud preparation, kNN normals and <= 30
    point-to-plane iterations each, one stream of launches for the batch).  Reports ICP iterations/sec."""
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
                "mean_fitness": float(np.mean([r["residual"] for r in res])),
                # per iteration and hypothesis N_src * 24 B + N_tgt * 48 B compulsory; a brute-force search is N_src * N_tgt * 8 flop
                "algorithmic_bytes": float(sum(r["iterations"] * (r["n_source"] * 24 + r["n_target"] * 48) for r in res if r["residual"] >= 0)),
                "brute_force_flops": float(sum(r["iterations"] * r["n_source"] * r["n_target"] * 8 for r in res if r["residual"] >= 0))}
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
                "templates": n,
                "points_source": [int(r["n_source"]) for r in res if r["status"] == 0],
                "points_target": [int(r["n_target"]) for r in res if r["status"] == 0],
                "iterations": [int(r["iterations"]) for r in res if r["status"] == 0]})
    return out


def host_cpu_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(frames, bank, n_templates, n_timed=10, n_warm=2):
    """The CPU side of the same matching step, single thread like the reference, on the host cores of this box, over `n_timed`
    frames of the stream after `n_warm` warm-up frames.  Two implementations are timed on identical inputs:
      * "reference" (the `value` when oracle/_ref is present): the reference's OWN Detector::match (LL.cpp:1702-1777 minus the
        quantisers: spread, computeResponseMaps, linearize, matchClass, std::sort + std::unique), compiled from /root/reference
        by oracle/Makefile with the reference's flags (-O3, SSE2) - oracle/_ref/libll_ref_sse2.so;
      * "port": oracle/match_oracle.c, the restatement of the same loops (pinned to the former record by record).
    The quantisation (numpy in the oracle, OpenCV in the reference) is timed separately and reported as `incl_quantisation`:
    numpy is slower than OpenCV would be, so the quantisation-exclusive figure is the one that cannot flatter the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    import ll_ref
    od = lo.OracleDetector(NFEAT[0], T_LEVELS)
    feat, offs, wh = bank
    pb = lo.PackedBank(n_templates, 2, feat, offs, wh)
    have_ref = ll_ref.available("sse2")
    times, qtimes, rtimes, cands = [], [], [], 0
    use = frames[:n_warm + n_timed]
    ref_equal = None
    for i, (rgb, dep) in enumerate(use):
        _, raw, st, pyr, tq, tmatch = oracle_matches(od, lo, pb, rgb, dep, THRESHOLD)
        quant = [(p[0], p[1]) for p in pyr]
        if have_ref:
            t0 = time.perf_counter()
            rr = ll_ref.match(quant, T_LEVELS, {"obj": pb}, THRESHOLD, ["obj"], pre_unique=True)
            tr = time.perf_counter() - t0
            if i == 0:   # same pre-unique records, in the reference's own emission order
                ref_equal = bool(len(rr) == len(raw) and all(np.array_equal(rr[a], raw[b]) for a, b in (("x", "x"), ("y", "y"), ("sim", "sim"), ("tid", "tid"))))
        if i >= n_warm:
            times.append(tmatch); qtimes.append(tq)
            cands += st["coarse_candidates"]
            if have_ref:
                rtimes.append(tr)
    lms = [[lo.build_linear_memories(p[0], T_LEVELS[l]), lo.build_linear_memories(p[1], T_LEVELS[l])] for l, p in enumerate(pyr)]
    sizes = [(p[0].shape[1], p[0].shape[0]) for p in pyr]
    ncores = os.cpu_count() or 1
    mt = []
    for _ in range(3):
        t0 = time.perf_counter()
        lo.match_bank_c(pb, lms, sizes, T_LEVELS, THRESHOLD, ncores)
        mt.append(time.perf_counter() - t0)
    t_mt = float(np.median(mt))
    sec_port, sec_q = float(np.median(times)), float(np.median(qtimes))
    rate = lambda sec: n_templates * (W * H / 1e6) / sec
    port = {"value": rate(sec_port), "unit": "templates*Mpx/s", "cores": 1, "seconds_per_frame": sec_port,
            "what": "oracle/match_oracle.c (SSE2/SSSE3 C port of LL.cpp:1026-1941), linear memories + coarse + local, median of %d frames" % len(times)}
    out = dict(port, kind="port")
    if have_ref:
        sec_ref = float(np.median(rtimes))
        out = {"value": rate(sec_ref), "unit": "templates*Mpx/s", "cores": 1, "kind": "reference", "seconds_per_frame": sec_ref,
               "reference_lines": {"value": rate(sec_ref), "seconds_per_frame": sec_ref, "min": min(rtimes), "max": max(rtimes),
                                   "library": "oracle/_ref/libll_ref_sse2.so = LL.cpp:1022-1658, 1694-1941 cut from the reference checkout at build time, "
                                              "-O3 (the reference's flags: SSE2 paths), Detector::match from the quantised maps on",
                                   "equals_port_records": ref_equal},
               "port": port}
    out.update({
        "sample": "%d frames (after %d warm-up frames) x %d templates (same bank and stream as the GPU run); single thread; "
                  "median %.3f s/frame; %.1f coarse candidates/template; quantisation excluded from `value`"
                  % (len(times), n_warm, n_templates, out["seconds_per_frame"], cands / len(times) / n_templates),
        "incl_quantisation": {"value": rate(out["seconds_per_frame"] + sec_q), "seconds_per_frame": out["seconds_per_frame"] + sec_q,
                              "quantisation_seconds": sec_q,
                              "note": "quantisation = the oracle's numpy restatement of the OpenCV calls (LL.cpp:350-505, 729-880), NOT the reference's "
                                      "OpenCV: an upper bound on the CPU time of a frame"},
        "host_cpu": host_cpu_name(), "host_cores": ncores,
        "all_cores_variant": {"threads": ncores, "value": rate(t_mt),
                              "note": "port, templates split across pthreads, match loops only (not what the reference does); median of 3"}})
    return out


def icp_cpu_baseline(hypotheses=4):
    """CPU side of extras.icp: the oracle's pose_refine (numpy restatement of LL.cpp:27-155 + Open3D's VoxelDownSample /
    EstimateNormals(KNN 30) / RegistrationICP point-to-plane, brute-force neighbours) on the first `hypotheses` of the same 16
    hypotheses, single thread.  The reference times this call at linemod_and_levelup_test.py:362-376.  Labelled numpy: Open3D's
    KD-tree + OpenMP would be faster, so a second figure replaces the neighbour searches of the ICP loop by scipy's cKDTree."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    import synth
    from scipy.spatial import cKDTree
    K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
    rng = np.random.default_rng(7)
    scene_model = synth.synth_model_depth(100)
    scene = np.where(scene_model > 0, scene_model + 4, 0).astype(np.uint16)
    scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
    R, t = np.eye(3, dtype=np.float32), np.array([0, 0, 1000], np.float32)
    its, sec, its_kd, sec_kd = 0, 0.0, 0, 0.0
    for h in range(16):
        md = synth.synth_model_depth(100 + (h % 4))
        ys, xs = np.nonzero(md)
        xy = (int(xs.min()) + int(rng.integers(-2, 3)), int(ys.min()) + int(rng.integers(-2, 3)))   # same draws as icp_bench
        if h >= hypotheses:
            continue
        t0 = time.perf_counter()
        r = lo.pose_refine(scene, md, K, K, R, t, xy[0], xy[1], scene_from_scene=True)
        sec += time.perf_counter() - t0
        its += r["iterations"]
        # the ICP loop alone with KD-tree neighbour searches (same clouds, normals and update rule)
        src, tgt, nrm, T = r["src"], r["tgt"], r["normals"], np.array(r["init_guess"])
        t0 = time.perf_counter()
        tree = cKDTree(tgt)
        pts = src @ T[:3, :3].T + T[:3, 3]
        prev = None
        for _ in range(lo.ICP_MAX_ITER + 1):
            d, j = tree.query(pts, k=1, distance_upper_bound=lo.ICP_MAX_DIST)
            ok = np.isfinite(d)
            n = int(ok.sum())
            cur = (n / len(pts), float(np.sqrt((d[ok] ** 2).sum() / n)) if n else 0.0)
            if prev is not None:
                its_kd += 1
                if abs(prev[0] - cur[0]) < lo.ICP_REL and abs(prev[1] - cur[1]) < lo.ICP_REL:
                    break
            prev = cur
            upd = np.eye(4)
            if n >= 6:
                p, q, nt = pts[ok], tgt[j[ok]], nrm[j[ok]]
                rres = ((p - q) * nt).sum(1)
                J = np.concatenate([np.cross(p, nt), nt], 1)
                try:
                    x = np.linalg.solve(J.T @ J, -(J.T @ rres))
                    if np.all(np.isfinite(x)):
                        upd = lo._rot_xyz(x)
                except np.linalg.LinAlgError:
                    pass
            pts = pts @ upd[:3, :3].T + upd[:3, 3]
        sec_kd += time.perf_counter() - t0
    return {"kind": "port", "cores": 1, "hypotheses": hypotheses, "iterations_total": its, "seconds": sec,
            "icp_iters_per_sec": its / sec if sec > 0 else 0.0,
            "what": "oracle pose_refine (numpy, brute-force neighbours), whole call: clouds + voxel grid + normals + ICP; parity unpinned (Open3D absent)",
            "kdtree_icp_loop_only": {"iterations_total": its_kd, "seconds": sec_kd, "icp_iters_per_sec": its_kd / sec_kd if sec_kd > 0 else 0.0,
                                     "what": "the ICP evaluations alone (clouds, normals given) with scipy cKDTree searches + numpy normal equations: "
                                             "the kind of loop Open3D runs (KD-tree), one thread; not bit-compared"}}


def real_fixture_leg(device, steps=50, target=2000):
    """the reference's own detect_test (linemodLevelup/test.cpp:111-130) - fixture frame 0000 read as BGR,
    Detector(127, {5, 8}), bank `127` (89 template pyramids at 1000 mm), threshold 75 - with the bank tiled to ~2k pyramids
    (copies under new template ids) so that it is the configs[1] size.  Parity-checked against the CPU oracle (all pyramids),
    then the same live-stream loop as the headline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import linemod_oracle as lo
    import linemodLevelup_pybind as lm
    from helpers import GOLDEN, load_bgr, load_u16
    rgb, dep = load_bgr("0000_rgb.png"), load_u16("0000_dep.png")
    od = lo.OracleDetector(127, [5, 8])
    od.readClasses(["06_template"], os.path.join(GOLDEN, "bank127_%s.yaml.gz"))
    pyrs = od.class_templates["06_template"]
    reps = max(1, (target + len(pyrs) - 1) // len(pyrs))
    pb = lo.pack_bank(pyrs * reps, 2)
    det = lm.Detector(127, [5, 8], device=device)
    det.addClassPacked("06_template", pb.feat, pb.tmpl_off, pb.tmpl_wh)
    classes = ["06_template"]
    od2 = lo.OracleDetector(127, [5, 8])
    want, _, st, _, _, tmatch = oracle_matches(od2, lo, pb, rgb, dep, THRESHOLD)
    got = det.matchArray([rgb, dep], THRESHOLD, classes)
    equal = same_records(got, want)
    frames = [(rgb.copy(), dep.copy()) for _ in range(4)]
    dt, tm = pipelined_host_stream(det, frames, classes, THRESHOLD, steps)
    n = pb.num_pyramids
    return {"workload": "test.cpp:111-130 detect_test: fixture frame 0000 (BGR) x bank 127 tiled x%d = %d template pyramids, Detector(127,{5,8}), "
                        "threshold 75, 640x480" % (reps, n),
            "templates": n, "ms_per_frame": dt * 1e3, "value": n * (W * H / 1e6) / dt, "unit": "templates*Mpx/s", "steps": steps, **stream_shape(det, tm),
            "coarse_candidates": tm.get("coarse_candidates"), "matches_pre_unique": tm.get("matches_pre_unique"), "matches_final": int(len(got)),
            "coarse_ms": tm.get("coarse_ms"), "local_ms": tm.get("local_ms"), "frontend_ms": tm.get("frontend_ms"),
            "equals_oracle": bool(equal), "top_match": ({"x": int(got[0]["x"]), "y": int(got[0]["y"]), "similarity": float(got[0]["similarity"]),
                                                        "template_id": int(got[0]["template_id"])} if len(got) else None),
            "cpu_port_seconds_per_frame": tmatch,
            "note": "T = {5, 8} is the geometry of every reference fixture bank; GT bbox origin of the object is (331, 130) (test.cpp:86)"}


if __name__ == "__main__":
    main()
