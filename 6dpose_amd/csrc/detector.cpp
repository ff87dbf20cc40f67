// libamdlinemod.so — host orchestration + C ABI (include/amd_linemod.h) of the MI355X LINE-MOD
// detector.  Mirrors linemodLevelup::Detector (LL.cpp:1663-2146): bank bookkeeping and the greedy
// template extraction on the host, every per-pixel / per-template stage in HIP kernels
// (frontend.hip, match.hip).  No CPU fallback: creation fails without a HIP device.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <system_error>
#include <chrono>
#include <memory>
#include <string>
#include <vector>
#include <map>

#include "detector_internal.h"
#include <sched.h>
#include <ctype.h>
#include "render_internal.h"

// ---- errors -----------------------------------------------------------------------------------
static thread_local std::string g_err;
int lm_set_error(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
extern "C" const char* lm_last_error(void) { return g_err.c_str(); }
static void pool_stop(lm_detector* d);
extern "C" const char* lm_version(void) { return "amd-linemod 0.1 (gfx950)"; }
// Binds the calling thread to the CPUs next to `device` (its PCI function's local_cpulist in sysfs): pinned staging buffers are then
// allocated, filled and read by the copy engine on the GPU's own NUMA node.  On a two-socket host a process that happens to start on
// the far socket otherwise uploads every frame across the socket link (0.12 instead of 0.065 ms per VGA frame).
extern "C" int lm_bind_thread_near_device(int device, char* cpulist_out, size_t cap) {
    if (cpulist_out && cap) cpulist_out[0] = 0;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) {
        (void)hipGetLastError();
        return lm_set_error(LM_ERR_NO_DEVICE, "no PCI bus id for device %d", device);
    }
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return lm_set_error(LM_ERR_IO, "cannot read %s", path.c_str());
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return lm_set_error(LM_ERR_IO, "empty %s", path.c_str());
    cpu_set_t want, have;
    CPU_ZERO(&want);
    int ncpu = 0;
    for (const char* p = line; *p;) {                          // "0-47,96-143"
        if (*p < '0' || *p > '9') { ++p; continue; }
        char* e = nullptr;
        long a = strtol(p, &e, 10), b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &want); ++ncpu; }
        p = e;
    }
    if (ncpu == 0) return lm_set_error(LM_ERR_IO, "no CPUs listed in %s", path.c_str());
    if (sched_getaffinity(0, sizeof(have), &have) == 0) {       // never widen what the caller (cgroup, numactl, taskset) allowed
        cpu_set_t both;
        CPU_AND(&both, &want, &have);
        if (CPU_COUNT(&both) == 0) return lm_set_error(LM_ERR_INVALID, "none of the device's local CPUs (%s) is allowed for this thread", line);
        want = both;
    }
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return lm_set_error(LM_ERR_INVALID, "sched_setaffinity failed");
    if (cpulist_out && cap) {
        size_t n = strlen(line);
        while (n && (line[n - 1] == '\n' || line[n - 1] == ' ')) line[--n] = 0;
        snprintf(cpulist_out, cap, "%s", line);
    }
    return LM_OK;
}

extern "C" int lm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" void lm_free(void* p) { free(p); }

static int ensure_pinned(lm_detector* d, size_t bytes) {
    if (bytes <= d->pinned_bytes) return LM_OK;
    if (d->pinned) (void)hipHostFree(d->pinned);
    d->pinned = nullptr; d->pinned_bytes = 0;
    HIP_TRY(hipHostMalloc(&d->pinned, bytes, hipHostMallocDefault));
    d->pinned_bytes = bytes;
    return LM_OK;
}

// NORMAL_LUT plane (normal_lut.i): round(atan2(y-10, x-10)/45deg) mod 8, one-hot (z-independent)
static void make_normal_lut(uint8_t lut[400]) {
    const double PI = 3.14159265358979323846;
    for (int iy = 0; iy < 20; ++iy)
        for (int ix = 0; ix < 20; ++ix) {
            double ang = atan2((double)(iy - 10), (double)(ix - 10)) * 180.0 / PI;
            if (ang < 0) ang += 360.0;
            int lab = ((int)floor(ang / 45.0 + 0.5)) % 8;
            lut[iy * 20 + ix] = (uint8_t)(1u << lab);
        }
}

extern "C" int lm_detector_create(int num_features, const int* T, int num_levels, int device, lm_detector** out) {
    if (!out) return lm_set_error(LM_ERR_INVALID, "out is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return lm_set_error(LM_ERR_NO_DEVICE, "no HIP device visible (%s); libamdlinemod has no CPU fallback",
                            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return lm_set_error(LM_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    if (T && (num_levels < 1 || num_levels > kMaxLevels))
        return lm_set_error(LM_ERR_INVALID, "num_levels must be in 1..%d", kMaxLevels);
    lm_detector* d = new lm_detector();
    if (num_features > 0) d->num_features = num_features;
    if (T) {
        d->T_at_level.assign(T, T + num_levels);
        for (int t : d->T_at_level)
            if (t < 1) { delete d; return lm_set_error(LM_ERR_INVALID, "T must be >= 1"); }
    }
    d->pyramid_levels = (int)d->T_at_level.size();
    d->device = device;
    // Four streams: front end | coarse pass | refinement | duplicate removal + multi-GPU exchange.  Streams that share a hardware
    // queue run in submission order, so they must land on different queues.  The HIP runtime pools its hardware queues
    // (GPU_MAX_HW_QUEUES, 4 by default) PER PRIORITY and hands a new stream the least used queue of its pool; a process that
    // holds other streams (torch's default stream, RCCL's high-priority one) competes for the same pools.  Spread over all three:
    // front end and coarse pass HIGH (short kernels on the latency path of the next frame), the refinement — the one long kernel,
    // which fills whatever the short ones leave free — alone in the LOW pool, duplicate removal + exchange NORMAL.
    // Measured, ms/frame, stand-alone process / torch + RCCL process (world 1, device exchange):
    //   everything normal 0.223 / 0.34 (front end and matching serialised on one queue);   this assignment 0.223 / 0.222;
    //   coarse or front end at normal priority 0.223 / 0.230-0.237 (coarse shares a queue: half overlapped);
    //   exchange at low priority: its five dependent steps take ~50 us each and the frames in flight no longer hide the latency;
    //   GPU_MAX_HW_QUEUES=8: 0.223 / 0.40-0.50 (more queues than the hardware runs at once: they are time-sliced).
    // LM_STREAM_PRIO="f-mx" overrides (digits: 0 normal, 1 low, 2 high; position 0 = frame / front-end stream, 2 = matching stream, 3 = exchange
    // stream; position 1 belonged to the coarse stream that round 5 removed and is ignored).
    int prio_least = 0, prio_greatest = 0;
    if (hipSetDevice(device) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    int prio[4] = {prio_greatest, prio_greatest, prio_least, 0};
    if (const char* pe = getenv("LM_STREAM_PRIO"))
        for (int i = 0; i < 4 && pe[i]; ++i) prio[i] = pe[i] == '1' ? prio_least : (pe[i] == '2' ? prio_greatest : 0);
    bool streams_ok = hipSetDevice(device) == hipSuccess;
    if (streams_ok)
        streams_ok = hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, prio[0]) == hipSuccess &&
                     hipStreamCreateWithPriority(&d->mstream, hipStreamNonBlocking, prio[2]) == hipSuccess &&
                     hipStreamCreateWithPriority(&d->xchg.stream, hipStreamNonBlocking, prio[3]) == hipSuccess;
    if (!streams_ok) {
        delete d;
        return lm_set_error(LM_ERR_NO_DEVICE, "cannot initialise HIP device %d", device);
    }
    // the copy stream of the live-stream ingest too, now: every stream of the detector takes its hardware queue before anything created
    // later (torch, RCCL) does
    if (hipStreamCreateWithFlags(&d->ingest.stream, hipStreamNonBlocking) == hipSuccess)
        for (int i = 0; i < lm_detector::kSlots; ++i) { (void)hipEventCreate(&d->ingest.t0[i]); (void)hipEventCreate(&d->ingest.t1[i]); }
    else d->ingest.stream = nullptr;
    for (auto& ev : d->ev) (void)hipEventCreate(&ev);
    for (auto& sl : d->slot) {
        for (auto& e : sl.ev) (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);   // timing only: nobody synchronises on them, and a default record costs the queue a cache write-back + invalidate (5-6 us between two kernels; sl.done keeps the fence)
        (void)hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&sl.fe_done, hipEventDisableTiming);
    }
    d->work_cls = std::make_shared<std::vector<int32_t>>();
    d->work_tid = std::make_shared<std::vector<int32_t>>();
    if (const char* ff = getenv("LM_FE_FUSED")) d->fe_fused = ff[0] && ff[0] != '0';
    if (knobs().frame_batch > 0) d->batch_max = std::min(knobs().frame_batch, kMaxBatch);
    if (knobs().batch_queue > 0) d->keep_queued = knobs().batch_queue;
    if (knobs().launch_slack_us > 0) d->launch_slack_ms = knobs().launch_slack_us * 1e-3f;
    if (const char* ac = getenv("LM_ASYNC_COLLECT")) d->async_collect = ac[0] && ac[0] != '0';
    if (const char* ht = getenv("LM_HOST_THREADS")) d->pool.threads = std::max(0, std::min(8, atoi(ht)));
    if (const char* tl = getenv("LM_TILES")) d->use_tiles = tl[0] && tl[0] != '0';
    if (const char* ro = getenv("LM_REFERENCE_ORDER")) d->reference_order = ro[0] && ro[0] != '0';
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) d->num_cus = prop.multiProcessorCount;
    }
    uint8_t lut[400];
    make_normal_lut(lut);
    upload_normal_lut(lut);
    *out = d;
    return LM_OK;
}

extern "C" void lm_detector_destroy(lm_detector* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)lm_launch_pending(d);
    pool_stop(d);
    for (auto& sl : d->slot) { free(sl.prep); sl.prep = nullptr; }
    (void)hipStreamSynchronize(d->stream);
    if (d->mstream) (void)hipStreamSynchronize(d->mstream);
    d->frame_rgb.release(); d->frame_depth.release(); d->nrm_raw.release();
    for (int i = 0; i < lm_detector::kSlots; ++i) {
        if (d->ingest.pinned[i]) (void)hipHostFree(d->ingest.pinned[i]);
        d->ingest.d_rgb[i].release();
        if (d->ingest.t0[i]) (void)hipEventDestroy(d->ingest.t0[i]);
        if (d->ingest.t1[i]) (void)hipEventDestroy(d->ingest.t1[i]);
    }
    if (d->ingest.stream) (void)hipStreamDestroy(d->ingest.stream);
    for (int a = 0; a < lm_detector::kSlots; ++a) { d->lm_arena[a].release(); d->sm_arena[a].release(); }
    for (int a = 0; a < lm_detector::kSlots; ++a) { d->bits_arena[a].release(); d->cbits_arena[a].release(); }
    for (auto& b : d->slot_rgb) b.release();
    for (auto& b : d->slot_depth) b.release();
    for (auto& l : d->lvl) { l.rgb.release(); l.mag.release(); l.ang.release(); l.nrm.release(); l.mask[0].release(); l.mask[1].release(); }
    d->d_entries.release(); d->d_feat_off.release(); d->d_feat_xy.release(); d->d_feat_word.release(); d->d_run_mask.release(); d->d_work.release();
    d->d_cands.release(); d->d_counters.release(); d->d_final.release(); d->d_matches_dev.release(); d->d_hash.release(); d->d_distinct_keys.release(); d->d_tiles.release(); d->d_todo.release(); d->d_work_cls.release(); d->d_work_tid.release();
    for (int l = 0; l < kMaxLevels; ++l) { d->train.mask[l].release(); d->train.lab[l].release(); d->train.hrun[l].release(); }
    d->train.user_mask.release();
    d->train.keys.release(); d->train.counts.release(); d->train.bbox.release(); d->train.out.release();
    for (auto& sl : d->slot) {
        if (sl.h_matches) (void)hipHostFree(sl.h_matches);
        if (sl.h_distinct) (void)hipHostFree(sl.h_distinct);
        if (sl.h_counters) (void)hipHostFree(sl.h_counters);
        if (sl.fe_done) (void)hipEventDestroy(sl.fe_done);
        for (auto& e : sl.ev) if (e) (void)hipEventDestroy(e);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (d->xchg.stream) (void)hipStreamSynchronize(d->xchg.stream);
    for (int a = 0; a < lm_detector::kSlots; ++a) {
        d->xchg.d_merged[a].release();
        d->xchg.d_runs.release();
        if (d->xchg.h_merged[a]) (void)hipHostFree(d->xchg.h_merged[a]);
        if (d->xchg.done[a]) (void)hipEventDestroy(d->xchg.done[a]);
    }
    if (d->xchg.stream) (void)hipStreamDestroy(d->xchg.stream);
    if (d->pinned) (void)hipHostFree(d->pinned);
    for (auto& ev : d->ev) if (ev) (void)hipEventDestroy(ev);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->mstream) (void)hipStreamDestroy(d->mstream);
    delete d;
}

// ---- frame upload + front end --------------------------------------------------------------------
// Zero tail after the 8 labels of one (level, modality) block: covers the reference's reads past a
// phase row and the reads of padded / out-of-image features redirected to it, for any
// position offset < Wd*Hd plus one 16-row window.
static size_t lm_tail_pad(int Wd, int Hd) { return (size_t)Wd * Hd + (size_t)16 * Wd + 2048; }

// (Re)allocates per-level buffers and the LM arena for a W x H frame; validates the reference's
// preconditions (LL.cpp:1136, 1217-1218).
static int setup_geometry(lm_detector* d, int W, int H, bool check_match_preconditions) {
    const int L = d->pyramid_levels;
    FrameGeom g{};
    g.levels = L;
    int w = W, h = H;
    size_t arena = 0, sarena = 0;
    for (int l = 0; l < L; ++l) {
        if (l > 0) { w /= 2; h /= 2; }
        if (w < 1 || h < 1) return lm_set_error(LM_ERR_INVALID, "image too small for %d pyramid levels", L);
        int T = d->T_at_level[l];
        if (check_match_preconditions) {
            if (((long)w * h) % 16 != 0)
                return lm_set_error(LM_ERR_INVALID, "(src.rows * src.cols) %% 16 == 0 violated at level %d (%dx%d) [LL.cpp:1136]", l, w, h);
            if (h % T != 0 || w % T != 0)
                return lm_set_error(LM_ERR_INVALID, "response_map.rows/cols %% T == 0 violated at level %d (%dx%d, T=%d) [LL.cpp:1217-1218]", l, w, h, T);
        }
        LevelGeom& lv = g.lv[l];
        lv.W = w; lv.H = h; lv.T = T; lv.Wd = w / T; lv.Hd = h / T;
        size_t block = (size_t)8 * T * T * lv.Wd * lv.Hd + lm_tail_pad(lv.Wd, lv.Hd);
        block = (block + 255) & ~(size_t)255;
        d->lm_block_bytes[l] = block;
        for (int m = 0; m < 2; ++m) {
            if (arena + block > 0xFFFFFFFFull) return lm_set_error(LM_ERR_INVALID, "frame too large for the LM arena");
            lv.lm_off[m] = (uint32_t)arena;
            arena += block;
        }
        // strip-major copy for the refinement (levels below the top): [8 labels][T*T phases][NS strips][Hd rows][16 B]
        // per modality, then one all-zero plane (read by padded features) and slack for the second aligned dword.
        lv.NS = (lv.Wd + 15) / 16;
        lv.sm_off[0] = lv.sm_off[1] = 0;
        if (l < L - 1) {
            const size_t splane = (size_t)lv.NS * lv.Hd * 16;
            const size_t sblock = (size_t)8 * T * T * splane;
            if (sarena + 2 * sblock + 3 * splane + 4096 > 0xFFFFFFFFull) return lm_set_error(LM_ERR_INVALID, "frame too large for the strip arena");
            lv.sm_off[0] = (uint32_t)sarena;
            lv.sm_off[1] = (uint32_t)(sarena + sblock);
            sarena += 2 * sblock + 3 * splane + 4096;
            sarena = (sarena + 255) & ~(size_t)255;
        }
    }
    const size_t n0 = (size_t)W * H;
    int rc;
    if ((rc = d->frame_rgb.ensure(n0 * 3))) return rc;
    if ((rc = d->frame_depth.ensure(n0))) return rc;
    if ((rc = d->nrm_raw.ensure(n0))) return rc;

    for (int a = 0; a < lm_detector::kSlots; ++a) {
        const bool realloc_arena = arena > d->lm_arena[a].cap;
        if ((rc = d->lm_arena[a].ensure(arena))) return rc;
        if (realloc_arena || d->fW != W || d->fH != H)   // zero tails (and everything else) once
            HIP_TRY(hipMemsetAsync(d->lm_arena[a].p, 0, d->lm_arena[a].cap, d->stream));
        const bool realloc_sarena = std::max<size_t>(sarena, 256) > d->sm_arena[a].cap;
        if ((rc = d->sm_arena[a].ensure(std::max<size_t>(sarena, 256)))) return rc;
        if (realloc_sarena || d->fW != W || d->fH != H) HIP_TRY(hipMemsetAsync(d->sm_arena[a].p, 0, d->sm_arena[a].cap, d->stream));
        {   // pair stream of the top level's two blocks (zero tails included); written whole by every front end, so never cleared
            const LevelGeom& top = g.lv[L - 1];
            d->cbits_byte0 = top.lm_off[0] & ~31u;
            d->cbits_npairs = (uint32_t)((top.lm_off[1] + d->lm_block_bytes[L - 1] - d->cbits_byte0 + 31) / 32);
            const size_t cbytes = (size_t)d->cbits_npairs * 8 + 64;
            const bool realloc_cbits = cbytes > d->cbits_arena[a].cap;
            if ((rc = d->cbits_arena[a].ensure(cbytes))) return rc;
            if (realloc_cbits || d->fW != W || d->fH != H) HIP_TRY(hipMemsetAsync(d->cbits_arena[a].p, 0, d->cbits_arena[a].cap, d->stream));
        }
        {   // strip records: the strip arena's layout at half the offsets; its zero planes stay zero
            const size_t bbytes = std::max<size_t>(sarena, 256) / 2 + 64;
            const bool realloc_bits = bbytes > d->bits_arena[a].cap;
            if ((rc = d->bits_arena[a].ensure(bbytes))) return rc;
            if (realloc_bits || d->fW != W || d->fH != H) HIP_TRY(hipMemsetAsync(d->bits_arena[a].p, 0, d->bits_arena[a].cap, d->stream));
        }
    }
    for (int l = 0; l < L; ++l) {
        LevelBufs& b = d->lvl[l];
        b.W = g.lv[l].W; b.H = g.lv[l].H;
        size_t n = (size_t)b.W * b.H;
        if (l > 0 && (rc = b.rgb.ensure(n * 3))) return rc;
        if ((rc = b.mag.ensure(n))) return rc;
        if ((rc = b.ang.ensure(n))) return rc;
        if ((rc = b.nrm.ensure(n))) return rc;
    }
    d->geom = g;
    d->fW = W; d->fH = H;
    return LM_OK;
}

static int upload_frame(lm_detector* d, const uint8_t* rgb, const uint16_t* depth, int W, int H,
                        const uint8_t* const* masks, bool check_match_preconditions) {
    if (!rgb || !depth) return lm_set_error(LM_ERR_INVALID, "rgb/depth is null");
    if (W < 16 || H < 16 || W > 16384 || H > 16384) return lm_set_error(LM_ERR_INVALID, "unsupported frame size %dx%d", W, H);
    HIP_TRY(hipSetDevice(d->device));
    d->frame_valid = false;
    if (d->n_submitted != d->n_collected)   // the front end's buffers (and, on a size change, the arenas) belong to the frames in flight
        return lm_set_error(LM_ERR_INVALID, "frames in flight: collect them before uploading another frame this way (lm_detector_submit_frame streams)");
    LM_DIAG_IDLE(d, "upload_frame");
    int rc = setup_geometry(d, W, H, check_match_preconditions);
    if (rc) return rc;
    d->cur_rgb = d->frame_rgb.p; d->cur_depth = d->frame_depth.p;
    const size_t n = (size_t)W * H;
    const bool m0 = masks && masks[0], m1 = masks && masks[1];
    size_t bytes = n * 3 + n * 2 + (m0 ? n : 0) + (m1 ? n : 0);
    if ((rc = ensure_pinned(d, bytes))) return rc;
    uint8_t* st = (uint8_t*)d->pinned;
    memcpy(st, rgb, n * 3);
    memcpy(st + n * 3, depth, n * 2);
    HIP_TRY(hipEventRecord(d->ev[6], d->stream));
    HIP_TRY(hipMemcpyAsync(d->frame_rgb.p, st, n * 3, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->frame_depth.p, st + n * 3, n * 2, hipMemcpyHostToDevice, d->stream));
    size_t off = n * 5;
    for (int m = 0; m < 2; ++m) {
        d->have_mask[m] = masks && masks[m];
        if (!d->have_mask[m]) continue;
        memcpy(st + off, masks[m], n);
        if ((rc = d->lvl[0].mask[m].ensure(n))) return rc;
        HIP_TRY(hipMemcpyAsync(d->lvl[0].mask[m].p, st + off, n, hipMemcpyHostToDevice, d->stream));
        off += n;
        for (int l = 1; l < d->pyramid_levels; ++l) {       // resize(INTER_NEAREST), LL.cpp:573-578, 874-879
            const LevelBufs& a = d->lvl[l - 1];
            LevelBufs& b = d->lvl[l];
            if ((rc = b.mask[m].ensure((size_t)b.W * b.H))) return rc;
            launch_nn_down2(a.mask[m].p, b.mask[m].p, a.W, a.H, d->stream);
        }
    }
    HIP_TRY(hipEventRecord(d->ev[7], d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));   // staging buffer is reused by the next call
    (void)hipEventElapsedTime(&d->last_h2d_ms, d->ev[6], d->ev[7]);
    d->frame_valid = true;
    return LM_OK;
}

// quantise every level; build_lm=false for addTemplate (only the quantised maps are needed)
static int run_frontend(lm_detector* d, bool build_lm, int arena = 0, bool share_launches = true) {
    // One stream: measured on MI355X, forking the colour / pyramid / depth chains onto three streams
    // (events, also inside the hipGraph) cost more in cross-stream synchronisation (+26 us) than the
    // ~3 us kernels could overlap.  Instead the jobs that do not depend on each other can share a LAUNCH (k_fe_stage): per
    // level {colour chain, normals + median or their nearest-neighbour pyramid, pyrDown to the next level}, then the linear
    // memories of all levels — 7 -> 3 launches at two levels.  That is what a LONE frame and the training views get (launch
    // latency is their critical path: synchronous match 0.409 -> 0.390 ms).  The matching path (run_frontend_batch) uses the same
    // shared launches for all frames of a batch; this function serves addTemplate (build_lm = false).
    hipStream_t s = d->stream;
    const int L = d->pyramid_levels;
    const float thr_sq = d->weak_threshold * d->weak_threshold;
    const bool fused = d->fe_fused && share_launches;
    FeStage st{};
    for (int l = 0; l < L; ++l) {
        LevelBufs& b = d->lvl[l];
        const uint8_t* src = l == 0 ? d->cur_rgb : b.rgb.p;
        if (fused) {
            st.njobs = 0;
            fe_job_colour(st.job[st.njobs++], src, b.mag.p, b.ang.p, b.W, b.H, thr_sq);                                   // LL.cpp:367-504
            if (l == 0) fe_job_normals(st.job[st.njobs++], d->cur_depth, d->nrm_raw.p, b.nrm.p, b.W, b.H, d->distance_threshold,
                                       d->difference_threshold);                                                          // LL.cpp:729-819
            else fe_job_nn_down2(st.job[st.njobs++], d->lvl[l - 1].nrm.p, b.nrm.p, d->lvl[l - 1].W, d->lvl[l - 1].H);     // LL.cpp:857-880
            if (l + 1 < L) fe_job_pyrdown(st.job[st.njobs++], src, d->lvl[l + 1].rgb.p, b.W, b.H);                        // LL.cpp:557-581
            launch_fe_stage(st, s);
            continue;
        }
        if (l > 0) {
            const LevelBufs& a = d->lvl[l - 1];
            launch_pyrdown_rgb(l == 1 ? d->cur_rgb : a.rgb.p, b.rgb.p, a.W, a.H, s);   // LL.cpp:557-581
            launch_nn_down2(a.nrm.p, b.nrm.p, a.W, a.H, s);                                   // LL.cpp:857-880
        } else {
            launch_normals_fused(d->cur_depth, d->nrm_raw.p, b.nrm.p, b.W, b.H, d->distance_threshold,
                                 d->difference_threshold, s);                                 // LL.cpp:729-819
        }
        launch_color_quant(src, b.mag.p, b.ang.p, b.W, b.H, thr_sq, s);                       // LL.cpp:367-504
    }
    if (build_lm) {
        st.njobs = 0;
        for (int l = 0; l < L; ++l) {
            LevelBufs& b = d->lvl[l];
            const LevelGeom& lv = d->geom.lv[l];
            const bool strips = l < L - 1;
            const uint8_t* quant[2] = {b.ang.p, b.nrm.p};
            const uint8_t* mask[2] = {d->have_mask[0] ? b.mask[0].p : nullptr, d->have_mask[1] ? b.mask[1].p : nullptr};
            uint8_t* lmp[2] = {d->lm_arena[arena].p + lv.lm_off[0], d->lm_arena[arena].p + lv.lm_off[1]};
            uint8_t* smp[2] = {strips ? d->sm_arena[arena].p + lv.sm_off[0] : nullptr, strips ? d->sm_arena[arena].p + lv.sm_off[1] : nullptr};
            if (fused) fe_job_build_lm(st.job[st.njobs++], quant, mask, lmp, smp, b.W, b.H, lv.T);
            else launch_build_lm(quant, mask, lmp, smp, b.W, b.H, lv.T, s);
        }
        if (fused) launch_fe_stage(st, s);
        d->last_arena = arena;
    }
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

// ---- bank -----------------------------------------------------------------------------------------
static int validate_pyramid(const lm_detector* d, const TemplatePyramid& tp) {
    if ((int)tp.size() != d->pyramid_levels * 2)
        return lm_set_error(LM_ERR_INVALID, "template pyramid has %d entries, detector expects %d", (int)tp.size(),
                            d->pyramid_levels * 2);
    for (const Template& t : tp) {
        if (t.features.size() > 8191) return lm_set_error(LM_ERR_INVALID, "templ.features.size() <= 8191 [LL.cpp:1291]");
        for (const Feature& f : t.features) {
            if (f.label < 0 || f.label > 7) return lm_set_error(LM_ERR_INVALID, "feature label %d outside [0,8)", f.label);
            if (f.x < -32768 || f.x > 32767 || f.y < -32768 || f.y > 32767)
                return lm_set_error(LM_ERR_INVALID, "feature coordinate outside the supported int16 range");
        }
    }
    return LM_OK;
}

// Detector::addTemplate on the frame resident in frame_rgb / frame_depth (LL.cpp:1943-1975).
static int add_template_resident(lm_detector* d, const uint8_t* mask, int width, int height, const char* class_id) {
    // quantise() in addTemplate passes object_mask to every modality (LL.cpp:1957), but the masked
    // quantised image is not used by extractTemplate; only the unmasked maps + the mask are.
    int rc;
    if ((rc = run_frontend(d, false))) return rc;
    d->frame_valid = false;   // LM arena not built for this frame
    const int L = d->pyramid_levels;
    std::vector<TemplatePyramid>& tps = d->class_templates[class_id];   // created even on failure, LL.cpp:1947
    d->bank_dirty = true;
    TemplatePyramid tp((size_t)2 * L);
    std::vector<uint8_t> hmask, nmask;
    if (mask) hmask.assign(mask, mask + (size_t)width * height);
    size_t nf = (size_t)d->num_features;
    int ext = d->extract_threshold;
    std::vector<float> mag;
    std::vector<uint8_t> ang, nrm;
    for (int l = 0; l < L; ++l) {
        const LevelBufs& b = d->lvl[l];
        const size_t n = (size_t)b.W * b.H;
        if (l > 0) {
            nf /= 2;            // LL.cpp:560, 860
            ext /= 2;           // LL.cpp:861
            if (mask) {         // resize(mask, INTER_NEAREST)
                const LevelBufs& a = d->lvl[l - 1];
                nmask.resize(n);
                for (int y = 0; y < b.H; ++y)
                    for (int x = 0; x < b.W; ++x) nmask[(size_t)y * b.W + x] = hmask[(size_t)(2 * y) * a.W + 2 * x];
                hmask.swap(nmask);
            }
        }
        mag.resize(n); ang.resize(n); nrm.resize(n);
        HIP_TRY(hipMemcpyAsync(mag.data(), b.mag.p, n * sizeof(float), hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipMemcpyAsync(ang.data(), b.ang.p, n, hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipMemcpyAsync(nrm.data(), b.nrm.p, n, hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
        const uint8_t* mp = mask ? hmask.data() : nullptr;
        // reference order is modality-major (LL.cpp:1954-1968); the outcome (-1 on any failure) is the same
        if (!extract_color_template(mag.data(), ang.data(), mp, b.W, b.H, nf, d->strong_threshold, l, tp[2 * l])) return -1;
        if (!extract_normal_template(nrm.data(), mp, b.W, b.H, nf, ext, l, tp[2 * l + 1])) return -1;
    }
    crop_templates(tp);
    if ((rc = validate_pyramid(d, tp))) return rc;
    tps.push_back(std::move(tp));
    return (int)tps.size() - 1;
}

// The scratch of the device selection for `views` views of the current geometry, and the maps it reads (the detector's level buffers).
static int train_buffers(lm_detector* d, int views, TrainGeom& g) {
    lm_detector::Train& T = d->train;
    const int L = d->pyramid_levels;
    const size_t out_words = 4 + 3 * (size_t)std::max(1, d->num_features);
    int rc;
    g.levels = L;
    for (int l = 0; l < L; ++l) {
        const LevelBufs& b = d->lvl[l];
        const size_t nl = (size_t)b.W * b.H;
        if ((rc = T.mask[l].ensure(nl)) || (rc = T.lab[l].ensure(nl)) || (rc = T.hrun[l].ensure(nl))) return rc;
        g.W[l] = b.W; g.H[l] = b.H; g.mag[l] = b.mag.p; g.ang[l] = b.ang.p; g.nrm[l] = b.nrm.p;
        g.mask[l] = T.mask[l].p; g.lab[l] = T.lab[l].p; g.hrun[l] = T.hrun[l].p;
    }
    const size_t keys_view = (size_t)L * 2 * kTrainCap, counts_view = (size_t)L * 16;
    if ((rc = T.keys.ensure(keys_view * views)) || (rc = T.counts.ensure(counts_view * views)) || (rc = T.bbox.ensure(4 * (size_t)views)) ||
        (rc = T.out.ensure((size_t)views * L * 2 * out_words)))
        return rc;
    return LM_OK;
}

// One view's output of k_train_select ([levels][2][out_words], every status 1) as a template pyramid of the class: cropTemplates,
// the bank's limits, push_back.  Returns the template id.
static int push_selected_pyramid(lm_detector* d, std::vector<TemplatePyramid>& tps, const int32_t* out_view, size_t out_words) {
    const int L = d->pyramid_levels;
    TemplatePyramid tp((size_t)2 * L);
    for (int e = 0; e < 2 * L; ++e) {
        const int32_t* o = out_view + (size_t)e * out_words;
        Template& t = tp[e];
        t.pyramid_level = e / 2;
        t.features.resize((size_t)o[1]);
        for (int k = 0; k < o[1]; ++k) t.features[k] = Feature{o[4 + 3 * k], o[4 + 3 * k + 1], o[4 + 3 * k + 2]};
    }
    crop_templates(tp);
    int rc = validate_pyramid(d, tp);
    if (rc) return rc;
    tps.push_back(std::move(tp));
    return (int)tps.size() - 1;
}

// Detector::addTemplate with an object mask, selection on the device (train.hip): the quantised maps never leave HBM, only the
// chosen features come back.  Candidate lists beyond what the selection kernel sorts in LDS go to add_template_resident.
static int add_template_device(lm_detector* d, const uint8_t* mask, int width, int height, const char* class_id) {
    int rc;
    if ((rc = run_frontend(d, false))) return rc;
    d->frame_valid = false;   // LM arena not built for this frame
    lm_detector::Train& T = d->train;
    const int L = d->pyramid_levels;
    const int nf_cap = std::max(1, d->num_features);
    const size_t out_words = 4 + 3 * (size_t)nf_cap, npx = (size_t)width * height;
    TrainGeom g{};
    if ((rc = train_buffers(d, 1, g)) || (rc = T.user_mask.ensure(npx))) return rc;
    hipStream_t s = d->stream;
    HIP_TRY(hipMemcpyAsync(T.user_mask.p, mask, npx, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(T.counts.p, 0, (size_t)L * 16 * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(T.bbox.p, 0x80, 4 * sizeof(int32_t), s));
    launch_train_prep(d->frame_depth.p, T.user_mask.p, g, d->strong_threshold * d->strong_threshold, d->extract_threshold, T.keys.p, kTrainCap, T.counts.p,
                      T.bbox.p, s);
    if (launch_train_select(T.keys.p, T.counts.p, g, kTrainCap, d->num_features, nf_cap, 1, T.out.p, s))
        return lm_set_error(LM_ERR_HIP, "cannot reserve LDS for the selection kernel");
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h_out((size_t)L * 2 * out_words);
    HIP_TRY(hipMemcpyAsync(h_out.data(), T.out.p, h_out.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    bool ok = true, host_path = false;
    for (int e = 0; e < 2 * L; ++e) {
        const int32_t st = h_out[(size_t)e * out_words];
        host_path |= st == 2;
        ok &= st == 1;
    }
    if (host_path) return add_template_resident(d, mask, width, height, class_id);
    std::vector<TemplatePyramid>& tps = d->class_templates[class_id];   // created even on failure, LL.cpp:1947
    d->bank_dirty = true;
    if (!ok) return -1;
    return push_selected_pyramid(d, tps, h_out.data(), out_words);
}

extern "C" int lm_detector_add_template(lm_detector* d, const uint8_t* rgb, const uint16_t* depth, const uint8_t* mask,
                                        int width, int height, const char* class_id) {
    if (!d || !class_id) return lm_set_error(LM_ERR_INVALID, "null argument");
    int rc = upload_frame(d, rgb, depth, width, height, nullptr, false);
    if (rc) return rc;
    // with an object mask (what every training loop of the reference passes) the selection runs on the device; LM_TRAIN_HOST=1 and
    // detectors beyond kTrainMaxFeatures features keep it on the host, as does a call without mask (no erosion, candidates anywhere)
    const char* force_host = getenv("LM_TRAIN_HOST");
    bool on_device = mask && !(force_host && force_host[0] && force_host[0] != '0') && d->num_features >= 1 && d->num_features <= kTrainMaxFeatures;
    if (on_device) {          // the device works on object / background; a grey mask (cv::erode takes minima, cv::subtract differences) stays on the host
        uint8_t v = 0;
        const size_t npx = (size_t)width * height;
        for (size_t i = 0; i < npx && on_device; ++i)
            if (mask[i]) { if (!v) v = mask[i]; else on_device = mask[i] == v; }
    }
    return on_device ? add_template_device(d, mask, width, height, class_id) : add_template_resident(d, mask, width, height, class_id);
}

// render_train (linemod_and_levelup_test.py:170-252) on the device: the rendered colour / depth images go from the
// rasteriser's buffers into the detector's frame buffers without touching the host, the quantisers and the feature selection
// (train.hip) run there too, and only the chosen features, the bounding boxes and the candidate counts come back — once per
// chunk of views, not per view.  A view whose candidate lists exceed what the selection kernel sorts in LDS (very large
// objects), or a detector with more than kTrainMaxFeatures features, takes the host selection (add_template_resident), which
// yields the same templates; LM_TRAIN_HOST=1 forces it (tests compare the two).
static int add_rendered_view_host(lm_detector* d, lm_mesh* m, int i, int width, int height, const char* class_id, std::vector<uint16_t>& hdepth,
                                  std::vector<uint8_t>& hmask, int32_t* box_wh_view) {
    const size_t npx = (size_t)width * height;
    d->frame_valid = false;
    d->have_mask[0] = d->have_mask[1] = false;
    d->cur_rgb = d->frame_rgb.p; d->cur_depth = d->frame_depth.p;
    HIP_TRY(hipMemcpyAsync(d->frame_rgb.p, m->d_rgb + (size_t)i * npx * 3, npx * 3, hipMemcpyDeviceToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->frame_depth.p, m->d_depth + (size_t)i * npx, npx * 2, hipMemcpyDeviceToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(hdepth.data(), m->d_depth + (size_t)i * npx, npx * 2, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    int x0 = width, y0 = height, x1 = -1, y1 = -1;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const bool on = hdepth[(size_t)y * width + x] > 0;
            hmask[(size_t)y * width + x] = on ? 255 : 0;                       // mask = (depth > 0) * 255 (:238)
            if (on) { x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y); }
        }
    if (box_wh_view) {                                                         // xmax - xmin, ymax - ymin (:235-236)
        box_wh_view[0] = x1 >= 0 ? x1 - x0 : 0;
        box_wh_view[1] = y1 >= 0 ? y1 - y0 : 0;
    }
    if (x1 < 0) return -1;
    return add_template_resident(d, hmask.data(), width, height, class_id);
}

extern "C" int lm_detector_add_templates_rendered(lm_detector* d, lm_mesh* m, const char* class_id, int count, int width, int height,
                                                  const float* Ks, const float* Rs, const float* ts, float clip_near, float clip_far,
                                                  float ambient, int ssaa, int32_t* template_ids, int32_t* box_wh) {
    if (!d || !m || !class_id || count < 0 || (count && (!Ks || !Rs || !ts || !template_ids)))
        return lm_set_error(LM_ERR_INVALID, "null argument");
    if (m->device != d->device) return lm_set_error(LM_ERR_INVALID, "mesh and detector live on different devices");
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "a frame is in flight: collect it first");
    LM_DIAG_IDLE(d, "lm_detector_add_templates_rendered");
    if (width < 16 || height < 16) return lm_set_error(LM_ERR_INVALID, "unsupported frame size %dx%d", width, height);
    HIP_TRY(hipSetDevice(d->device));
    const size_t npx = (size_t)width * height;
    const size_t per_view = npx * (size_t)ssaa * ssaa * sizeof(unsigned long long);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)2 << 30) / std::max<size_t>(per_view, 1)));
    std::vector<uint16_t> hdepth(npx);
    std::vector<uint8_t> hmask(npx);
    const char* force_host = getenv("LM_TRAIN_HOST");
    const bool on_device = !(force_host && force_host[0] && force_host[0] != '0') && d->num_features >= 1 && d->num_features <= kTrainMaxFeatures;
    const int L = d->pyramid_levels;
    const int nf_cap = std::max(1, d->num_features);
    const size_t out_words = 4 + 3 * (size_t)nf_cap;
    std::vector<int32_t> h_out, h_bbox;
    int rc;
    for (int c0 = 0; c0 < count; c0 += chunk) {
        const int n = std::min(chunk, count - c0);
        if ((rc = lm_mesh_render_device(m, n, width, height, Ks + 9 * (size_t)c0, Rs + 9 * (size_t)c0, ts + 3 * (size_t)c0, clip_near, clip_far,
                                        ambient, ssaa, true, true)))
            return rc;
        HIP_TRY(hipStreamSynchronize(m->s));
        if ((rc = setup_geometry(d, width, height, false))) return rc;
        if (!on_device) {
            for (int i = 0; i < n; ++i) {
                const int id = add_rendered_view_host(d, m, i, width, height, class_id, hdepth, hmask, box_wh ? box_wh + 2 * ((size_t)c0 + i) : nullptr);
                if (id < -1) return id;
                template_ids[(size_t)c0 + i] = id;
            }
            continue;
        }
        // ---- device selection: prepare every view of the chunk, select them all in one launch ----
        lm_detector::Train& T = d->train;
        TrainGeom g{};
        if ((rc = train_buffers(d, n, g))) return rc;
        const size_t keys_view = (size_t)L * 2 * kTrainCap, counts_view = (size_t)L * 16;
        hipStream_t s = d->stream;
        HIP_TRY(hipMemsetAsync(T.counts.p, 0, counts_view * n * sizeof(uint32_t), s));
        HIP_TRY(hipMemsetAsync(T.bbox.p, 0x80, 4 * (size_t)n * sizeof(int32_t), s));        // large negative: k_train_mask takes maxima
        d->frame_valid = false;
        d->have_mask[0] = d->have_mask[1] = false;
        const float strong_sq = d->strong_threshold * d->strong_threshold;
        for (int i = 0; i < n; ++i) {
            d->cur_rgb = d->frame_rgb.p; d->cur_depth = d->frame_depth.p;
            HIP_TRY(hipMemcpyAsync(d->frame_rgb.p, m->d_rgb + (size_t)i * npx * 3, npx * 3, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->frame_depth.p, m->d_depth + (size_t)i * npx, npx * 2, hipMemcpyDeviceToDevice, s));
            if ((rc = run_frontend(d, false))) return rc;
            launch_train_prep(d->frame_depth.p, nullptr, g, strong_sq, d->extract_threshold, T.keys.p + keys_view * i, kTrainCap, T.counts.p + counts_view * i,
                              T.bbox.p + 4 * (size_t)i, s);
        }
        if (launch_train_select(T.keys.p, T.counts.p, g, kTrainCap, d->num_features, nf_cap, n, T.out.p, s))
            return lm_set_error(LM_ERR_HIP, "cannot reserve LDS for the selection kernel");
        HIP_TRY(hipGetLastError());
        h_out.resize((size_t)n * L * 2 * out_words);
        h_bbox.resize(4 * (size_t)n);
        HIP_TRY(hipMemcpyAsync(h_out.data(), T.out.p, h_out.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h_bbox.data(), T.bbox.p, h_bbox.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<TemplatePyramid>& tps = d->class_templates[class_id];   // created even when every view fails, LL.cpp:1947
        d->bank_dirty = true;
        for (int i = 0; i < n; ++i) {
            const int32_t* bb = &h_bbox[4 * (size_t)i];
            const bool any = bb[2] >= 0;
            int id = -1;
            bool host_path = false;
            if (any) {
                bool ok = true;
                for (int e = 0; e < 2 * L; ++e) {
                    const int32_t st = h_out[((size_t)i * L * 2 + e) * out_words];
                    host_path |= st == 2;
                    ok &= st == 1;
                }
                if (host_path) {                                               // this view through the host selection, in view order
                    id = add_rendered_view_host(d, m, i, width, height, class_id, hdepth, hmask, nullptr);
                    if (id < -1) return id;
                } else if (ok) {
                    if ((id = push_selected_pyramid(d, tps, &h_out[(size_t)i * L * 2 * out_words], out_words)) < -1) return id;
                }
            }
            if (box_wh) {                                                      // xmax - xmin, ymax - ymin (:235-236)
                box_wh[2 * ((size_t)c0 + i)] = any ? bb[2] + bb[0] : 0;
                box_wh[2 * ((size_t)c0 + i) + 1] = any ? bb[3] + bb[1] : 0;
            }
            template_ids[(size_t)c0 + i] = id;
        }
    }
    return LM_OK;
}

extern "C" int lm_detector_read_class(lm_detector* d, const char* path, const char* class_id_override) {
    if (!d || !path) return lm_set_error(LM_ERR_INVALID, "null argument");
    std::string cid, err;
    std::vector<std::string> mods;
    int levels = 0;
    std::vector<TemplatePyramid> tps;
    if (!read_class_yaml(path, cid, mods, levels, tps, err)) {
        bool assertion = err.find("LL.cpp") != std::string::npos;
        return lm_set_error(assertion ? LM_ERR_INVALID : LM_ERR_IO, "%s", err.c_str());
    }
    if (mods.size() != 2 || mods[0] != "ColorGradient" || mods[1] != "DepthNormal")
        return lm_set_error(LM_ERR_INVALID, "modalities mismatch [LL.cpp:2047-2051]");
    if (levels != d->pyramid_levels)
        return lm_set_error(LM_ERR_INVALID, "(int)fn[\"pyramid_levels\"] == pyramid_levels violated (%d vs %d) [LL.cpp:2052]",
                            levels, d->pyramid_levels);
    if (class_id_override && class_id_override[0]) cid = class_id_override;
    else if (d->class_templates.count(cid))
        return lm_set_error(LM_ERR_INVALID, "class '%s' already present [LL.cpp:2059]", cid.c_str());
    for (const TemplatePyramid& tp : tps) { int rc = validate_pyramid(d, tp); if (rc) return rc; }
    if (!d->class_templates.count(cid)) d->class_templates[cid] = std::move(tps);   // map::insert keeps an existing key
    d->bank_dirty = true;
    return LM_OK;
}

extern "C" int lm_detector_write_class(lm_detector* d, const char* class_id, const char* path) {
    if (!d || !class_id || !path) return lm_set_error(LM_ERR_INVALID, "null argument");
    auto it = d->class_templates.find(class_id);
    if (it == d->class_templates.end()) return lm_set_error(LM_ERR_NOT_FOUND, "unknown class '%s' [LL.cpp:2096]", class_id);
    std::string err;
    if (!write_class_yaml(path, it->first, it->second, d->pyramid_levels, err)) return lm_set_error(LM_ERR_IO, "%s", err.c_str());
    return LM_OK;
}

// Detector::write / Detector::read (LL.cpp:2013-2041) with the modality parameters of ColorGradient::write (:686-692) and
// DepthNormal::write (:1012-1020), OpenCV FileStorage YAML 1.0 layout.  read() clears the classes like the reference.
extern "C" int lm_detector_write_params(const lm_detector* d, const char* path) {
    if (!d || !path) return lm_set_error(LM_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "w");
    if (!f) return lm_set_error(LM_ERR_IO, "cannot open for writing: %s", path);
    auto real = [](float v) {                                   // cv::FileStorage prints 10.f as "10."
        char b[64];
        snprintf(b, sizeof(b), "%.8g", (double)v);
        std::string s(b);
        if (s.find_first_of(".eEn") == std::string::npos) s += ".";
        return s;
    };
    fprintf(f, "%%YAML:1.0\n---\npyramid_levels: %d\nT: [", d->pyramid_levels);
    for (size_t i = 0; i < d->T_at_level.size(); ++i) fprintf(f, "%s %d", i ? "," : "", d->T_at_level[i]);
    fprintf(f, " ]\nmodalities:\n");
    fprintf(f, "   -\n      type: ColorGradient\n      weak_threshold: %s\n      num_features: %d\n      strong_threshold: %s\n",
            real(d->weak_threshold).c_str(), d->num_features, real(d->strong_threshold).c_str());
    fprintf(f, "   -\n      type: DepthNormal\n      distance_threshold: %d\n      difference_threshold: %d\n      num_features: %d\n"
               "      extract_threshold: %d\n",
            d->distance_threshold, d->difference_threshold, d->num_features, d->extract_threshold);
    if (fclose(f) != 0) return lm_set_error(LM_ERR_IO, "write failed: %s", path);
    return LM_OK;
}

extern "C" int lm_detector_read_params(lm_detector* d, const char* path) {
    if (!d || !path) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "a frame is in flight: collect it first");
    FILE* f = fopen(path, "r");
    if (!f) return lm_set_error(LM_ERR_IO, "cannot open: %s", path);
    int levels = -1, nf[2] = {-1, -1}, dist = d->distance_threshold, diff = d->difference_threshold, ext = d->extract_threshold;
    float weak = d->weak_threshold, strong = d->strong_threshold;
    std::vector<int> T;
    std::vector<std::string> types;
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
        char* p = line;
        while (*p == ' ' || *p == '\t' || *p == '-') ++p;
        char key[64];
        if (sscanf(p, "%63[A-Za-z_]:", key) != 1) continue;
        const char* v = strchr(p, ':') + 1;
        const std::string k(key);
        const std::string cur = types.empty() ? "" : types.back();
        if (k == "pyramid_levels") levels = atoi(v);
        else if (k == "T") { for (const char* q = v; *q; ++q) if (*q >= '0' && *q <= '9') { T.push_back(atoi(q)); while (*q >= '0' && *q <= '9') ++q; --q; } }
        else if (k == "type") { char t[64] = {0}; sscanf(v, " %63s", t); types.push_back(t); }
        else if (k == "weak_threshold") weak = (float)atof(v);
        else if (k == "strong_threshold") strong = (float)atof(v);
        else if (k == "num_features") { if (cur == "ColorGradient") nf[0] = atoi(v); else if (cur == "DepthNormal") nf[1] = atoi(v); }
        else if (k == "distance_threshold") dist = atoi(v);
        else if (k == "difference_threshold") diff = atoi(v);
        else if (k == "extract_threshold") ext = atoi(v);
    }
    fclose(f);
    if (levels < 1 || levels > kMaxLevels || (int)T.size() != levels) return lm_set_error(LM_ERR_IO, "%s: pyramid_levels / T missing or inconsistent", path);
    if (types.size() != 2 || types[0] != "ColorGradient" || types[1] != "DepthNormal")   // Modality::create (LL.cpp:320-328) knows these two
        return lm_set_error(LM_ERR_INVALID, "%s: modalities must be [ColorGradient, DepthNormal]", path);
    if (nf[0] <= 0 || nf[0] != nf[1]) return lm_set_error(LM_ERR_INVALID, "%s: the modalities must agree on num_features (one bank layout)", path);
    for (int t : T) if (t < 1) return lm_set_error(LM_ERR_INVALID, "T must be >= 1");
    d->class_templates.clear();                                   // LL.cpp:2015
    d->bank_dirty = true; d->work_valid = false; d->frame_valid = false;
    d->pyramid_levels = levels; d->T_at_level = T;
    d->num_features = nf[0]; d->weak_threshold = weak; d->strong_threshold = strong;
    d->distance_threshold = dist; d->difference_threshold = diff; d->extract_threshold = ext;
    d->fW = d->fH = 0;                                            // geometry depends on T: rebuilt by the next frame
    return LM_OK;
}

extern "C" int lm_detector_add_class_packed(lm_detector* d, const char* class_id, int num_pyramids, const int32_t* features,
                                            const int32_t* tmpl_offsets, const int32_t* tmpl_wh) {
    if (!d || !class_id || num_pyramids < 0 || (num_pyramids && (!features || !tmpl_offsets || !tmpl_wh)))
        return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (d->class_templates.count(class_id)) return lm_set_error(LM_ERR_INVALID, "class '%s' already present", class_id);
    const int E = d->pyramid_levels * 2;
    std::vector<TemplatePyramid> tps((size_t)num_pyramids);
    for (int p = 0; p < num_pyramids; ++p) {
        TemplatePyramid& tp = tps[p];
        tp.resize(E);
        for (int e = 0; e < E; ++e) {
            size_t k = (size_t)p * E + e;
            Template& t = tp[e];
            t.width = tmpl_wh[2 * k]; t.height = tmpl_wh[2 * k + 1]; t.pyramid_level = e / 2;
            int a = tmpl_offsets[k], b = tmpl_offsets[k + 1];
            if (a < 0 || b < a) return lm_set_error(LM_ERR_INVALID, "tmpl_offsets not monotone");
            t.features.resize((size_t)(b - a));
            for (int i = a; i < b; ++i) t.features[i - a] = Feature{features[3 * (size_t)i], features[3 * (size_t)i + 1], features[3 * (size_t)i + 2]};
        }
        int rc = validate_pyramid(d, tp);
        if (rc) return rc;
    }
    d->class_templates[class_id] = std::move(tps);
    d->bank_dirty = true;
    return LM_OK;
}

extern "C" int lm_detector_num_classes(const lm_detector* d) { return d ? (int)d->class_templates.size() : 0; }
extern "C" const char* lm_detector_class_id(const lm_detector* d, int index) {
    if (!d || index < 0 || index >= (int)d->class_templates.size()) return nullptr;
    auto it = d->class_templates.begin();
    std::advance(it, index);
    return it->first.c_str();
}
extern "C" int lm_detector_num_templates(const lm_detector* d, const char* class_id) {
    if (!d) return 0;
    if (!class_id) { int n = 0; for (auto& kv : d->class_templates) n += (int)kv.second.size(); return n; }
    auto it = d->class_templates.find(class_id);
    return it == d->class_templates.end() ? 0 : (int)it->second.size();
}
extern "C" int lm_detector_pyramid_levels(const lm_detector* d) { return d ? d->pyramid_levels : 0; }
extern "C" int lm_detector_get_T(const lm_detector* d, int level) {
    return (d && level >= 0 && level < d->pyramid_levels) ? d->T_at_level[level] : -1;
}

extern "C" int lm_detector_get_template(const lm_detector* d, const char* class_id, int template_id, int index, int32_t* width,
                                        int32_t* height, int32_t* pyramid_level, int32_t* num_features, int32_t* features,
                                        int capacity) {
    if (!d || !class_id) return lm_set_error(LM_ERR_INVALID, "null argument");
    auto it = d->class_templates.find(class_id);
    if (it == d->class_templates.end()) return lm_set_error(LM_ERR_NOT_FOUND, "unknown class '%s' [LL.cpp:1979]", class_id);
    if (template_id < 0 || (size_t)template_id >= it->second.size())
        return lm_set_error(LM_ERR_INVALID, "template_id out of range [LL.cpp:1980]");
    const TemplatePyramid& tp = it->second[template_id];
    if (index < 0 || index >= (int)tp.size()) return lm_set_error(LM_ERR_INVALID, "template index out of range");
    const Template& t = tp[index];
    if (width) *width = t.width;
    if (height) *height = t.height;
    if (pyramid_level) *pyramid_level = t.pyramid_level;
    if (num_features) *num_features = (int32_t)t.features.size();
    if (features)
        for (int i = 0; i < capacity && i < (int)t.features.size(); ++i) {
            features[3 * i] = t.features[i].x; features[3 * i + 1] = t.features[i].y; features[3 * i + 2] = t.features[i].label;
        }
    return LM_OK;
}

extern "C" int lm_detector_set_shard(lm_detector* d, int rank, int world) {
    if (!d || world < 1 || rank < 0 || rank >= world) return lm_set_error(LM_ERR_INVALID, "bad shard (%d of %d)", rank, world);
    d->shard_rank = rank; d->shard_world = world;
    return LM_OK;
}

static inline int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

// Flatten the bank for the current frame geometry and upload it: TemplEntry per (pyramid, level);
// per feature the byte offset of its linear-memory run from the arena start (accessLinearMemory,
// LL.cpp:1248-1271; floor division so that window offsets that are multiples of T stay exact) and
// packed int16 x,y.  Entries are padded to a multiple of kFeatBatch with features that read the
// level's zero tail; at the top level features outside the image (LL.cpp:1330) are redirected there too.
static int upload_bank(lm_detector* d) {
    const int L = d->pyramid_levels;
    d->bank_classes.clear(); d->bank_class_base.clear(); d->bank_class_count.clear();
    d->h_entries.clear();
    d->work_valid = false;
    std::vector<int32_t> off;
    std::vector<uint32_t> xy;
    std::vector<uint32_t> word, rmask;          // levels below the top: feat_word per feature, run_mask per 8 features (lm_kernels.h)
    const uint32_t pad_xy = 0x80008000u;   // x = y = -32768: never inside an image
    int flat = 0;
    for (auto& kv : d->class_templates) {
        d->bank_classes.push_back(kv.first);
        d->bank_class_base.push_back(flat);
        d->bank_class_count.push_back((int)kv.second.size());
        for (const TemplatePyramid& tp : kv.second) {
            for (int l = 0; l < L; ++l) {
                const LevelGeom& lv = d->geom.lv[l];
                const long npos = (long)lv.Wd * lv.Hd;
                const long zero_off = (long)lv.lm_off[1] + (long)8 * lv.T * lv.T * npos;   // tail of the normal block
                const long splane = (long)lv.NS * lv.Hd * 16;
                const uint32_t szero = (uint32_t)((long)lv.sm_off[1] + (long)8 * lv.T * lv.T * splane);   // the all-zero strip plane
                TemplEntry e{};
                e.feat_start = (uint32_t)off.size();
                const size_t n0 = tp[2 * l].features.size(), n1 = tp[2 * l + 1].features.size();
                e.nf = (uint16_t)(n0 + n1);
                e.width = tp[2 * l].width;      // matchClass uses tp[start] (first modality) for the clamp,
                e.height = tp[2 * l].height;    // similarity() each template's own size: checked equal below
                if (tp[2 * l + 1].width != e.width || tp[2 * l + 1].height != e.height)
                    return lm_set_error(LM_ERR_INVALID, "modalities of one pyramid level disagree on width/height");
                int mnx = 32767, mny = 32767, mxx = -32768, mxy = -32768;
                struct Rec { int32_t off; uint32_t xy; uint32_t base0; int cls; };
                std::vector<Rec> recs;
                const bool top = (l == L - 1);
                const long zero16 = (zero_off + 15) & ~15L;      // 16-aligned start of the zero tail
                for (int m = 0; m < 2; ++m)
                    for (const Feature& f : tp[2 * l + m].features) {
                        const int T = lv.T;
                        const int gx = f.x - floordiv(f.x, T) * T, gy = f.y - floordiv(f.y, T) * T;   // floor modulo
                        long o = (long)lv.lm_off[m] + ((long)f.label * T * T + (gy * T + gx)) * npos + (long)floordiv(f.y, T) * lv.Wd +
                                 floordiv(f.x, T);
                        const bool inside = f.x >= 0 && f.x < lv.W && f.y >= 0 && f.y < lv.H;
                        if (top && !inside) o = zero16;                                 // LL.cpp:1330
                        if (o < -(1L << 31) || o >= (1L << 31)) return lm_set_error(LM_ERR_INVALID, "feature offset overflow");
                        Rec r{};
                        r.off = (int32_t)o;
                        r.xy = (uint32_t)(uint16_t)(int16_t)f.x | ((uint32_t)(uint16_t)(int16_t)f.y << 16);
                        r.base0 = szero;
                        if (!top && f.x >= 0 && f.y >= 0) {   // only read on the fast path, where x, y >= 0: the 16-byte row of the feature's own cell
                            const long lx = f.x / T, ly = f.y / T;
                            r.base0 = (uint32_t)((long)lv.sm_off[m] + ((long)f.label * T * T + (gy * T + gx)) * splane + ((lx >> 4) * lv.Hd + ly) * 16);
                        }
                        // alignment class: byte phase of the run start (top level: flat offset; below: plane column)
                        r.cls = top ? (int)(o & 15) : (f.x >= 0 ? (f.x / T) & 15 : 0);
                        recs.push_back(r);
                        mnx = std::min(mnx, f.x); mny = std::min(mny, f.y); mxx = std::max(mxx, f.x); mxy = std::max(mxy, f.y);
                    }
                if (e.nf == 0) mnx = mny = mxx = mxy = 0;
                e.min_x = (int16_t)mnx; e.min_y = (int16_t)mny; e.max_x = (int16_t)mxx; e.max_y = (int16_t)mxy;
                std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.cls < b.cls; });
                std::vector<uint8_t> starts;             // per feature of this entry: 1 = first of a class run
                auto push_feat = [&](int32_t o, uint32_t pxy, uint32_t base0, int cls, bool start) {
                    off.push_back(o); xy.push_back(pxy); word.push_back((base0 & ~15u) | (uint32_t)cls); starts.push_back(start ? 1 : 0);
                };
                auto push_pad = [&](int cls) {           // a feature that reads zeros, in alignment class `cls`
                    push_feat((int32_t)(zero16 + (top ? cls : 0)), pad_xy, szero, cls, false);
                };
                int last_cls = 0;
                for (size_t i = 0; i < recs.size();) {
                    size_t j = i;
                    while (j < recs.size() && recs[j].cls == recs[i].cls) ++j;
                    // a run: <= kRunMax (even) features of one class, so that the packed-byte sums of the refinement cannot overflow
                    for (size_t k = i; k < j; ++k) push_feat(recs[k].off, recs[k].xy, recs[k].base0, recs[k].cls, (k - i) % kRunMax == 0);
                    last_cls = recs[i].cls;
                    if (!top && ((j - i) & 1)) push_pad(last_cls);   // the refinement consumes features in same-class pairs
                    i = j;
                }
                while ((off.size() - e.feat_start) % kFeatBatch) push_pad(last_cls);
                e.nf_padded = (uint16_t)(off.size() - e.feat_start);
                for (size_t k = 0; k < starts.size(); k += kFeatBatch) {           // kFeatBatch == 8: one mask word per batch
                    uint32_t mk = 0;
                    for (int u = 0; u < kFeatBatch; ++u) mk |= (uint32_t)starts[k + u] << u;
                    rmask.push_back(mk);
                }
                d->h_entries.push_back(e);
            }
            ++flat;
        }
    }
    int rc;
    if ((rc = d->d_entries.ensure(std::max<size_t>(1, d->h_entries.size())))) return rc;
    if ((rc = d->d_feat_off.ensure(std::max<size_t>(1, off.size())))) return rc;
    if ((rc = d->d_feat_xy.ensure(std::max<size_t>(1, xy.size())))) return rc;
    if ((rc = d->d_feat_word.ensure(std::max<size_t>(1, word.size())))) return rc;
    if ((rc = d->d_run_mask.ensure(std::max<size_t>(1, rmask.size())))) return rc;
    if (!d->h_entries.empty())
        HIP_TRY(hipMemcpy(d->d_entries.p, d->h_entries.data(), d->h_entries.size() * sizeof(TemplEntry), hipMemcpyHostToDevice));
    if (!off.empty()) {
        HIP_TRY(hipMemcpy(d->d_feat_off.p, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->d_feat_xy.p, xy.data(), xy.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->d_feat_word.p, word.data(), word.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->d_run_mask.p, rmask.data(), rmask.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // counter widths of the bit-plane kernels: the largest entry at the top level (k_coarse_bits) and below it (k_local_bits)
    d->bits_max_nf = 0; d->cbits_max_nf = 0;
    for (size_t i = 0; i < d->h_entries.size(); ++i) {
        int& mx = (int)(i % (size_t)L) == L - 1 ? d->cbits_max_nf : d->bits_max_nf;
        mx = std::max(mx, (int)d->h_entries[i].nf);
    }
    // Does EVERY candidate of this bank have its windows inside their planes at every level below the top (k_local's `all_in`)?  The
    // refinement clamps the window origin to x in [8T, W - width - 8T] (LL.cpp:1871-1880), so with that interval non-empty, gx = x / T - 8 >= 0
    // and (max_x + gx T) / T + 16 <= (max_x + W - width - 16 T) / T + 16 <= W / T whenever max_x <= width (W is a multiple of T); the same in
    // y.  Then k_local_bits leaves nothing for k_local's per-candidate path and the second launch is skipped.
    d->bits_all_in = true;
    for (size_t i = 0; i < d->h_entries.size(); ++i) {
        const int l = (int)(i % (size_t)L);
        if (l == L - 1) continue;
        const LevelGeom& lv = d->geom.lv[l];
        const TemplEntry& e = d->h_entries[i];
        d->bits_all_in = d->bits_all_in && e.min_x >= 0 && e.min_y >= 0 && e.max_x <= e.width && e.max_y <= e.height &&
                         lv.W - e.width - 16 * lv.T >= 0 && lv.H - e.height - 16 * lv.T >= 0 && lv.W % lv.T == 0 && lv.H % lv.T == 0;
    }
    d->bank_dirty = false;
    d->bank_geom_W = d->fW; d->bank_geom_H = d->fH;
    return LM_OK;
}

// ---- canonical merge (LL.cpp:1771-1776 with the total order of SURVEY A12) ---------------------------
static bool match_less(const lm_match& a, const lm_match& b) {
    if (a.similarity != b.similarity) return a.similarity > b.similarity;
    if (a.template_id != b.template_id) return a.template_id < b.template_id;
    if (a.class_index != b.class_index) return a.class_index < b.class_index;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
}
static bool match_eq(const lm_match& a, const lm_match& b) {   // Match::operator== (LL.h:243-246)
    return a.x == b.x && a.y == b.y && a.similarity == b.similarity && a.class_index == b.class_index;
}
// LSD radix sort on the 112-bit key (~similarity bits, template_id | class, y, x), 11-bit digits,
// digits that are constant over the input are skipped.  Equivalent to std::sort(match_less).
// distinct_input: the records hold no exact duplicates (k_dedupe removed them on the device): the hash pass is skipped.
// Scratch buffers are per thread and reused (a frame's list is merged in ~20 us; six allocations were a third of it).
static size_t merge_matches_impl(lm_match* m, size_t n, bool distinct_input) {
    if (!m || n == 0) return 0;
    if (n < 64) {
        std::sort(m, m + n, match_less);
        return (size_t)(std::unique(m, m + n, match_eq) - m);
    }
    struct Key { uint64_t hi, lo; };
    static thread_local std::vector<Key> keys;
    static thread_local std::vector<uint32_t> idx, tmp, table;
    static thread_local std::vector<lm_match> out;
    keys.resize(n);
    bool radix_ok = true;
    for (size_t i = 0; i < n; ++i) {
        uint32_t sb;
        memcpy(&sb, &m[i].similarity, 4);
        if ((sb >> 31) || m[i].similarity != m[i].similarity || m[i].template_id < 0 || m[i].class_index < 0 || m[i].class_index > 0xFFFF ||
            m[i].x < -32768 || m[i].x > 32767 || m[i].y < -32768 || m[i].y > 32767) { radix_ok = false; break; }
        if (sb == 0x80000000u) sb = 0;
        keys[i].hi = ((uint64_t)(~sb) << 32) | (uint32_t)m[i].template_id;
        keys[i].lo = ((uint64_t)m[i].class_index << 32) | ((uint64_t)(uint16_t)(m[i].y + 32768) << 16) | (uint16_t)(m[i].x + 32768);
    }
    if (!radix_ok) {   // negative / NaN similarities or out-of-range fields: comparison sort
        std::sort(m, m + n, match_less);
        return (size_t)(std::unique(m, m + n, match_eq) - m);
    }
    idx.clear();
    idx.reserve(n);
    if (distinct_input) {
        for (size_t i = 0; i < n; ++i) idx.push_back((uint32_t)i);
    } else {
        // Exact duplicates (same x, y, similarity, class AND template: several coarse candidates of one
        // template refined to the same position) are adjacent in the canonical order and removed by the
        // unique step anyway: drop them first with an open-addressing hash so that the sort sees ~n/5.
        size_t cap = 64;
        while (cap < 2 * n) cap <<= 1;
        table.assign(cap, 0xFFFFFFFFu);
        for (size_t i = 0; i < n; ++i) {
            uint64_t h = (keys[i].hi * 0x9E3779B97F4A7C15ull) ^ (keys[i].lo * 0xC2B2AE3D27D4EB4Full);
            size_t slot = (size_t)(h ^ (h >> 29)) & (cap - 1);
            for (;;) {
                uint32_t j = table[slot];
                if (j == 0xFFFFFFFFu) { table[slot] = (uint32_t)i; idx.push_back((uint32_t)i); break; }
                if (keys[j].hi == keys[i].hi && keys[j].lo == keys[i].lo) break;
                slot = (slot + 1) & (cap - 1);
            }
        }
    }
    const size_t nu = idx.size();
    tmp.resize(nu);
    constexpr int BITS = 11, NB = 1 << BITS;
    uint32_t hist[NB];
    // which bits vary at all: digits whose bits are constant over the input need no pass (and no histogram)
    uint64_t or_lo = 0, and_lo = ~0ull, or_hi = 0, and_hi = ~0ull;
    for (size_t i = 0; i < nu; ++i) { const Key& k = keys[idx[i]]; or_lo |= k.lo; and_lo &= k.lo; or_hi |= k.hi; and_hi &= k.hi; }
    const uint64_t var_lo = or_lo ^ and_lo, var_hi = or_hi ^ and_hi;
    for (int word = 0; word < 2; ++word)          // lo word first (least significant)
        for (int shift = 0; shift < (word == 0 ? 48 : 64); shift += BITS) {
            if ((((word == 0 ? var_lo : var_hi) >> shift) & (NB - 1)) == 0) continue;   // constant digit
            memset(hist, 0, sizeof(hist));
            for (size_t i = 0; i < nu; ++i) {
                uint64_t k = word == 0 ? keys[idx[i]].lo : keys[idx[i]].hi;
                ++hist[(k >> shift) & (NB - 1)];
            }
            uint32_t sum = 0;
            for (int b = 0; b < NB; ++b) { uint32_t c = hist[b]; hist[b] = sum; sum += c; }
            for (size_t i = 0; i < nu; ++i) {
                uint32_t id = idx[i];
                uint64_t k = word == 0 ? keys[id].lo : keys[id].hi;
                tmp[hist[(k >> shift) & (NB - 1)]++] = id;
            }
            idx.swap(tmp);
        }
    out.clear();
    out.reserve(nu);
    for (size_t i = 0; i < nu; ++i) {
        const lm_match& c = m[idx[i]];
        if (out.empty() || !match_eq(out.back(), c)) out.push_back(c);
    }
    memcpy(m, out.data(), out.size() * sizeof(lm_match));
    return out.size();
}
extern "C" size_t lm_merge_matches(lm_match* m, size_t n) { return merge_matches_impl(m, n, false); }

// numpy nms of the driver (linemod_and_levelup_test.py:34-61)
extern "C" int lm_nms_boxes(const double* boxes, const double* scores, int n, double thresh, int32_t* keep) {
    if (n <= 0 || !boxes || !scores || !keep) return 0;
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[i] = i;
    // scores.argsort()[::-1]: ascending stable-ish sort reversed -> among equal scores higher index first
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] < scores[b]; });
    std::reverse(order.begin(), order.end());
    std::vector<char> dead((size_t)n, 0);
    int kept = 0;
    for (int oi = 0; oi < n; ++oi) {
        int i = order[oi];
        if (dead[i]) continue;
        keep[kept++] = i;
        double ai = (boxes[4 * i + 2] - boxes[4 * i] + 1) * (boxes[4 * i + 3] - boxes[4 * i + 1] + 1);
        for (int oj = oi + 1; oj < n; ++oj) {
            int j = order[oj];
            if (dead[j]) continue;
            double xx1 = std::max(boxes[4 * i], boxes[4 * j]), yy1 = std::max(boxes[4 * i + 1], boxes[4 * j + 1]);
            double xx2 = std::min(boxes[4 * i + 2], boxes[4 * j + 2]), yy2 = std::min(boxes[4 * i + 3], boxes[4 * j + 3]);
            double w = std::max(0.0, xx2 - xx1 + 1), h = std::max(0.0, yy2 - yy1 + 1);
            double inter = w * h;
            double aj = (boxes[4 * j + 2] - boxes[4 * j] + 1) * (boxes[4 * j + 3] - boxes[4 * j + 1] + 1);
            double ovr = inter / (ai + aj - inter);
            if (!(ovr <= thresh)) dead[j] = 1;
        }
    }
    return kept;
}

// Translation NMS over refined poses (linemod_ros/detect.py:41-51, `nms_norms(ts, ts_scores, 40.0)` at :128): visit by
// score descending, keep, drop every later pose whose translation is within `thresh` of it (kept iff ||t_i - t_j|| > thresh,
// double precision, numpy's sqrt(dx*dx + dy*dy + dz*dz)).  Visiting order among EQUAL scores: the higher index first — what
// `scores.argsort()[::-1]` gives for n <= 16 (numpy's default argsort is an introsort: insertion sort, hence stable, up to 16
// elements; beyond that numpy's tie order is an implementation detail and this function's rule is this library's definition, not a
// reference-exact one).
extern "C" int lm_nms_norms(const double* ts, const double* scores, int n, double thresh, int32_t* keep) {
    if (n <= 0 || !ts || !scores || !keep) return 0;
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] < scores[b]; });
    std::reverse(order.begin(), order.end());
    std::vector<char> dead((size_t)n, 0);
    int kept = 0;
    for (int oi = 0; oi < n; ++oi) {
        const int i = order[oi];
        if (dead[i]) continue;
        keep[kept++] = i;
        for (int oj = oi + 1; oj < n; ++oj) {
            const int j = order[oj];
            if (dead[j]) continue;
            const double dx = ts[3 * i] - ts[3 * j], dy = ts[3 * i + 1] - ts[3 * j + 1], dz = ts[3 * i + 2] - ts[3 * j + 2];
            const double norm = sqrt(dx * dx + dy * dy + dz * dz);
            if (!(norm > thresh)) dead[j] = 1;
        }
    }
    return kept;
}

// cv::dnn::NMSBoxes(std::vector<Rect>, scores, score_threshold, nms_threshold, indices, eta, top_k) as linemodLevelup/test.cpp:
// 132-144 uses it (40x40 boxes at the match positions, score_threshold 0, nms_threshold 0.4).  OpenCV is un-vendored and its
// version unpinned; this follows the published algorithm of OpenCV 3.4's dnn/src/nms.inl.hpp (NMSFast_): candidates with
// score > score_threshold, std::stable_sort by score descending (ties keep input order), optional top_k cut, then greedily keep
// a box iff its overlap with every box kept so far is <= the adaptive threshold (which shrinks by eta after each keep while
// > 0.5 and eta < 1).  overlap = 1.f - float(jaccardDistance(a, b)), jaccardDistance in double on integer rectangle areas,
// 0 when both are empty.  rects: [n][4] int32 x, y, width, height.
extern "C" int lm_nms_boxes_cv(const int32_t* rects, const float* scores, int n, float score_threshold, float nms_threshold, float eta,
                               int top_k, int32_t* keep) {
    if (n <= 0 || !rects || !scores || !keep) return 0;
    std::vector<int> order;
    for (int i = 0; i < n; ++i) if (scores[i] > score_threshold) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
    if (top_k > 0 && (size_t)top_k < order.size()) order.resize((size_t)top_k);
    auto overlap = [&](int a, int b) -> float {
        const int32_t* A = rects + 4 * a; const int32_t* B = rects + 4 * b;
        const double Aa = (double)A[2] * A[3], Ab = (double)B[2] * B[3];
        if ((Aa + Ab) <= 2.220446049250313e-16) return 1.f - 0.f;                       // jaccardDistance: "identical": distance 0
        const int x1 = std::max(A[0], B[0]), y1 = std::max(A[1], B[1]);
        const int x2 = std::min(A[0] + A[2], B[0] + B[2]), y2 = std::min(A[1] + A[3], B[1] + B[3]);
        const double Aab = (x2 > x1 && y2 > y1) ? (double)(x2 - x1) * (y2 - y1) : 0.0;     // (a & b).area(): empty unless both extents positive
        return 1.f - (float)(1.0 - Aab / (Aa + Ab - Aab));
    };
    float adaptive = nms_threshold;
    int kept = 0;
    for (int idx : order) {
        bool ok = true;
        for (int k = 0; k < kept && ok; ++k) ok = overlap(idx, keep[k]) <= adaptive;
        if (ok) keep[kept++] = idx;
        if (ok && eta < 1.f && adaptive > 0.5f) adaptive *= eta;
    }
    return kept;
}

// ---- match ----------------------------------------------------------------------------------------
extern "C" int lm_detector_set_frame(lm_detector* d, const uint8_t* rgb, const uint16_t* depth, int width, int height,
                                     const uint8_t* const* masks) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    return upload_frame(d, rgb, depth, width, height, masks, true);
}

extern "C" int lm_detector_store_frame(lm_detector* d, int slot, const uint8_t* rgb, const uint16_t* depth, int width, int height) {
    if (!d || !rgb || !depth || slot < 0 || slot > 4095) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (width < 16 || height < 16 || width > 16384 || height > 16384) return lm_set_error(LM_ERR_INVALID, "unsupported frame size %dx%d", width, height);
    HIP_TRY(hipSetDevice(d->device));
    if ((size_t)slot >= d->slot_rgb.size()) {
        d->slot_rgb.resize(slot + 1); d->slot_depth.resize(slot + 1);
        d->slot_w.resize(slot + 1, 0); d->slot_h.resize(slot + 1, 0);
    }
    const size_t n = (size_t)width * height;
    int rc;
    if (d->slot_rgb[slot].cap < n * 3 || d->slot_depth[slot].cap < n)       // about to be reallocated: a frame in flight may still be
        HIP_TRY(hipStreamSynchronize(d->stream));                          // copying out of the old buffer
    if ((rc = d->slot_rgb[slot].ensure(n * 3))) return rc;
    if ((rc = d->slot_depth[slot].ensure(n))) return rc;
    // staged through the detector's pinned buffer like every other upload (a pageable hipMemcpy stages internally, chunk by chunk)
    if ((rc = ensure_pinned(d, n * 5))) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));                              // the staging buffer is shared with upload_frame
    uint8_t* st = (uint8_t*)d->pinned;
    memcpy(st, rgb, n * 3);
    memcpy(st + n * 3, depth, n * 2);
    HIP_TRY(hipMemcpyAsync(d->slot_rgb[slot].p, st, n * 3, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->slot_depth[slot].p, st + n * 3, n * 2, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->slot_w[slot] = width; d->slot_h[slot] = height;
    return LM_OK;
}

extern "C" int lm_detector_select_frame(lm_detector* d, int slot) {
    if (!d || slot < 0 || (size_t)slot >= d->slot_rgb.size() || d->slot_w[slot] <= 0)
        return lm_set_error(LM_ERR_INVALID, "no frame stored in slot %d", slot);
    HIP_TRY(hipSetDevice(d->device));
    const int W = d->slot_w[slot], H = d->slot_h[slot];
    if (W != d->fW || H != d->fH || d->lm_arena[0].cap == 0) {
        if (d->n_submitted != d->n_collected)   // setup_geometry reallocates and clears the arenas the frames in flight are reading
            return lm_set_error(LM_ERR_INVALID, "frame size changes (%dx%d -> %dx%d) with frames in flight: collect them first", d->fW, d->fH, W, H);
        d->frame_valid = false;
        LM_DIAG_IDLE(d, "lm_detector_select_frame (geometry change)");
        int rc = setup_geometry(d, W, H, true);
        if (rc) return rc;
    }
    d->frame_valid = false;
    const size_t n = (size_t)W * H;
    d->cur_rgb = d->frame_rgb.p; d->cur_depth = d->frame_depth.p;
    if (d->resident_reader) {                                 // a front end in flight (on the matching stream) may still read the resident frame
        HIP_TRY(hipStreamWaitEvent(d->stream, d->resident_reader, 0));
        d->resident_reader = nullptr;
    }
    HIP_TRY(hipMemcpyAsync(d->frame_rgb.p, d->slot_rgb[slot].p, n * 3, hipMemcpyDeviceToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->frame_depth.p, d->slot_depth[slot].p, n * 2, hipMemcpyDeviceToDevice, d->stream));
    d->have_mask[0] = d->have_mask[1] = false;
    d->last_h2d_ms = 0.f;
    d->frame_valid = true;
    return LM_OK;
}

static int build_work(lm_detector* d, const char* const* class_ids, int num_class_ids) {
    std::vector<std::string> key;
    if (class_ids && num_class_ids > 0)
        for (int i = 0; i < num_class_ids; ++i) key.push_back(class_ids[i] ? class_ids[i] : "");
    if (d->work_valid && key == d->work_key && d->work_key_rank == d->shard_rank && d->work_key_world == d->shard_world)
        return LM_OK;   // same selection as the previous call: the device-resident work list is reused
    d->work_pyr.clear();
    d->work_cls = std::make_shared<std::vector<int32_t>>();    // in-flight slots keep the old vectors alive
    d->work_tid = std::make_shared<std::vector<int32_t>>();
    std::vector<int> order;   // bank class index per position (-1 unknown)
    if (key.empty()) {
        for (size_t i = 0; i < d->bank_classes.size(); ++i) order.push_back((int)i);   // std::map order, LL.cpp:1756
    } else {
        for (const std::string& c : key) {
            int found = -1;
            for (size_t k = 0; k < d->bank_classes.size(); ++k)
                if (d->bank_classes[k] == c) { found = (int)k; break; }
            order.push_back(found);   // unknown classes are skipped, LL.cpp:1765-1767
        }
    }
    for (size_t pos = 0; pos < order.size(); ++pos) {
        int k = order[pos];
        if (k < 0) continue;
        for (int t = 0; t < d->bank_class_count[k]; ++t) {
            d->work_pyr.push_back(d->bank_class_base[k] + t);
            d->work_cls->push_back((int)pos);
            d->work_tid->push_back(t);
        }
    }
    // contiguous shard of the work list; template ids stay global
    const long N = (long)d->work_pyr.size();
    const long a = N * d->shard_rank / d->shard_world, b = N * (d->shard_rank + 1) / d->shard_world;
    if (d->shard_world > 1) {
        d->work_pyr = std::vector<int32_t>(d->work_pyr.begin() + a, d->work_pyr.begin() + b);
        *d->work_cls = std::vector<int32_t>(d->work_cls->begin() + a, d->work_cls->begin() + b);
        *d->work_tid = std::vector<int32_t>(d->work_tid->begin() + a, d->work_tid->begin() + b);
    }
    // frames in flight still read the device-resident work list: let them finish before it is replaced
    if (d->n_submitted != d->n_collected) {
        HIP_TRY(hipStreamSynchronize(d->mstream));
        if (d->xchg.stream) HIP_TRY(hipStreamSynchronize(d->xchg.stream));
    }
    int rc = d->d_work.ensure(std::max<size_t>(1, d->work_pyr.size()));
    if (rc) return rc;
    if ((rc = d->d_work_cls.ensure(std::max<size_t>(1, d->work_pyr.size())))) return rc;
    if ((rc = d->d_work_tid.ensure(std::max<size_t>(1, d->work_pyr.size())))) return rc;
    if (!d->work_pyr.empty()) {
        HIP_TRY(hipMemcpy(d->d_work.p, d->work_pyr.data(), d->work_pyr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->d_work_cls.p, d->work_cls->data(), d->work_pyr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->d_work_tid.p, d->work_tid->data(), d->work_pyr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // algorithmic bytes of the coarse pass over this work list: sum_m nfeat_m * template_positions
    {
        const int L = d->pyramid_levels;
        const LevelGeom& lv = d->geom.lv[L - 1];
        int64_t bytes = 0;
        for (int32_t p : d->work_pyr) {
            const TemplEntry& e = d->h_entries[(size_t)p * L + (L - 1)];
            int wf = (e.width - 1) / lv.T + 1, hf = (e.height - 1) / lv.T + 1;
            long tp = (long)(lv.Hd - hf) * lv.Wd + (lv.Wd - wf) + 1;
            if (tp > 0) bytes += (int64_t)e.nf * tp;
        }
        d->work_coarse_bytes = bytes;
    }
    d->work_key = key; d->work_key_rank = d->shard_rank; d->work_key_world = d->shard_world;
    d->work_valid = true;
    return LM_OK;
}

static int ensure_slot_buffers(lm_detector* d, lm_detector::Slot& sl, uint32_t match_cap) {
    if (!sl.h_counters)
        HIP_TRY(hipHostMalloc((void**)&sl.h_counters, 8 * sizeof(unsigned long long), hipHostMallocDefault));
    if (match_cap > sl.match_cap) {
        if (sl.h_matches) (void)hipHostFree(sl.h_matches);
        sl.h_matches = nullptr; sl.match_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&sl.h_matches, (size_t)match_cap * sizeof(Candidate), hipHostMallocDefault));
        if (sl.h_distinct) (void)hipHostFree(sl.h_distinct);
        sl.h_distinct = nullptr;
        HIP_TRY(hipHostMalloc((void**)&sl.h_distinct, (size_t)match_cap * sizeof(Candidate), hipHostMallocDefault));
        sl.match_cap = match_cap;
    }
    return LM_OK;
}

static int sync_all_streams(lm_detector* d) {
    HIP_TRY(hipStreamSynchronize(d->stream)); HIP_TRY(hipStreamSynchronize(d->mstream));
    if (d->xchg.stream) HIP_TRY(hipStreamSynchronize(d->xchg.stream));
    return LM_OK;
}

static bool tiles_wanted(const lm_detector* d) { return d->use_tiles && d->refine_mode != 2; }   // LM_TILES=0 / lm_detector_set_paths(2, .): every candidate on its own

// Grid of the refinement kernel for a batch of nb frames.  Per-candidate path (LM_TILES=0): 3 workgroups (12 waves) per CU — alone it
// is as fast as with every wave slot taken (it is bound by the vector L1, not by latency), and the free slots let the coarse pass of
// the next frame and the front end run beside it.  With tiles the work items are fewer and larger (a tile = two singles' worth of
// loads; ~5k items per 2k templates): a grid with more waves than items gives every wave at most one item and lets the hardware's
// workgroup dispatch do the balancing — 171 us (3 per CU, items dealt round-robin, slowest wave 2 tiles + 1 single) -> 122 (8) ->
// 103 (16 and more), profiles/r02_sweep_local_blocks.txt.  A batch has nb times the items: the grid grows with it.
static int local_grid(lm_detector* d, int nb) {
    if (knobs().local_blocks > 0) return knobs().local_blocks;
    return d->num_cus * (tiles_wanted(d) ? 16 : 3) * std::max(1, std::min(nb, 4));
}

// Grid of k_local_bits (a wave serves 8 candidates, ~2k groups per frame at configs[1]): the workgroups the chip holds at once (4 waves per SIMD = 4 workgroups of 256 per CU), whatever the batch: the waves stride over
// the items.  (Round 4 launched four times as many for batches of four and more frames; one wave per item and a dispatcher that has to place 4096
// workgroups cost 3-4 %: 204-207 -> 197-199 us per 8-frame launch, profiles/r05_local_sharing/kernel_times_grid_sweep.txt.)
static int bits_grid(lm_detector* d, int nb) {
    (void)nb;
    if (knobs().local_blocks > 0) return knobs().local_blocks;
    return d->num_cus * 4;
}

// Front end of a batch: the same three stages a lone frame takes (k_fe_stage: {colour chain, normals + median or their
// nearest-neighbour pyramid, pyrDown to the next level} per level, then the linear memories of every level), every stage ONE launch
// that carries the jobs of all frames of the batch.  Frame b keeps its intermediates in level_bufs(b, l) and writes the arenas of its
// own result slot.  7 launches per frame (round 2's per-slot graph) -> 3 per batch.
// direct_low / direct_top: nothing will read the byte planes of the levels below the top / of the top level — the bit planes are written
// straight from the quantised maps by one k_fe_bits launch at the end (frontend.hip) and the byte planes not at all.
#ifdef LM_DIAG
struct LaunchClock {                                          // LM_LAUNCH_PROF: host time of individual HIP calls of a batch launch
    const char* name; double t0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    explicit LaunchClock(const char* n) : name(n), t0(now()) {}
    ~LaunchClock() {
        static std::map<std::string, std::pair<double, long>> acc;
        auto& a = acc[name]; a.first += now() - t0; ++a.second;
        if (getenv("LM_LAUNCH_PROF") && a.second % 64 == 0) fprintf(stderr, "  call %-22s %.1f us (n=%ld)\n", name, 1e6 * a.first / a.second, a.second);
    }
};
#define LM_CLOCK(n) LaunchClock lm_clock_##__LINE__(n)
#else
#define LM_CLOCK(n)
#endif
static int run_frontend_batch(lm_detector* d, int first, int nb, hipStream_t s, bool direct_low, bool direct_top) {
    const int L = d->pyramid_levels;
    const float thr_sq = d->weak_threshold * d->weak_threshold;
    int rc;
    for (int b = 1; b < nb; ++b) {                       // intermediates of the batch's further frames (frame 0: setup_geometry)
        for (int l = 0; l < L; ++l) {
            LevelBufs& B = d->level_bufs(b, l);
            B.W = d->lvl[l].W; B.H = d->lvl[l].H;
            const size_t n = (size_t)B.W * B.H;
            if (l > 0 && (rc = B.rgb.ensure(n * 3))) return rc;
            if ((rc = B.mag.ensure(n))) return rc;
            if ((rc = B.ang.ensure(n))) return rc;
            if ((rc = B.nrm.ensure(n))) return rc;
        }
        if ((rc = d->nrm_raw_x[b - 1].ensure((size_t)d->fW * d->fH))) return rc;
    }
    FeStage st{};
    auto flush = [&]() { if (st.njobs) { LM_CLOCK("launch_fe_stage"); launch_fe_stage(st, s); } st.njobs = 0; };
    auto room = [&](int jobs) { if (st.njobs + jobs > kFeMaxJobs) flush(); };
    auto build_lm_jobs = [&](int l) {                        // linear memories of level l of every frame (its quantised maps are complete)
        if (l < L - 1 ? direct_low : direct_top) return;    // bit planes only: fe_bits_jobs below
        for (int b = 0; b < nb; ++b) {
            const int arena = (first + b) % lm_detector::kSlots;
            const lm_detector::Slot& sl = d->slot[arena];
            LevelBufs& B = d->level_bufs(b, l);
            const LevelGeom& lv = d->geom.lv[l];
            const bool strips = l < L - 1;
            const uint8_t* quant[2] = {B.ang.p, B.nrm.p};
            const uint8_t* mask[2] = {sl.have_mask[0] ? d->lvl[l].mask[0].p : nullptr, sl.have_mask[1] ? d->lvl[l].mask[1].p : nullptr};
            uint8_t* lmp[2] = {d->lm_arena[arena].p + lv.lm_off[0], d->lm_arena[arena].p + lv.lm_off[1]};
            uint8_t* smp[2] = {strips ? d->sm_arena[arena].p + lv.sm_off[0] : nullptr, strips ? d->sm_arena[arena].p + lv.sm_off[1] : nullptr};
            room(1);
            fe_job_build_lm(st.job[st.njobs++], quant, mask, lmp, smp, B.W, B.H, lv.T);
        }
    };
    // Launch l quantises level l of every frame and — beside it, they only need level l - 1 — builds the linear memories of level
    // l - 1; a last launch builds those of the top level.  (The memories of level 0 are 3/4 of that work: they no longer wait for
    // the quantisation of the small levels, and the last launch is a quarter of what it was.)
    for (int l = 0; l < L; ++l) {
        st.njobs = 0;
        for (int b = 0; b < nb; ++b) {
            const lm_detector::Slot& sl = d->slot[(first + b) % lm_detector::kSlots];
            LevelBufs& B = d->level_bufs(b, l);
            const uint8_t* src = l == 0 ? sl.in_rgb : B.rgb.p;
            room(3);
            fe_job_colour(st.job[st.njobs++], src, nullptr /* magnitudes: addTemplate only */, B.ang.p, B.W, B.H, thr_sq);      // LL.cpp:367-504
            if (l == 0) fe_job_normals(st.job[st.njobs++], sl.in_depth, b == 0 ? d->nrm_raw.p : d->nrm_raw_x[b - 1].p, B.nrm.p, B.W, B.H,
                                       d->distance_threshold, d->difference_threshold);                                          // LL.cpp:729-819
            else fe_job_nn_down2(st.job[st.njobs++], d->level_bufs(b, l - 1).nrm.p, B.nrm.p, d->level_bufs(b, l - 1).W, d->level_bufs(b, l - 1).H);   // LL.cpp:857-880
            if (l + 1 < L) fe_job_pyrdown(st.job[st.njobs++], src, d->level_bufs(b, l + 1).rgb.p, B.W, B.H);                     // LL.cpp:557-581
        }
        if (l > 0) build_lm_jobs(l - 1);
        flush();
    }
    build_lm_jobs(L - 1);
    flush();
    if (direct_low || direct_top) {                      // the bit planes of every frame in one launch
        st.njobs = 0;
        auto flush_bits = [&]() { if (st.njobs) { LM_CLOCK("launch_fe_bits"); launch_fe_bits(st, s); } st.njobs = 0; };
        for (int b = 0; b < nb; ++b) {
            const int arena = (first + b) % lm_detector::kSlots;
            const lm_detector::Slot& sl = d->slot[arena];
            for (int l = 0; l < L; ++l) {
                const bool top = l == L - 1;
                if (top ? !direct_top : !direct_low) continue;
                LevelBufs& B = d->level_bufs(b, l);
                const LevelGeom& lv = d->geom.lv[l];
                const uint8_t* quant[2] = {B.ang.p, B.nrm.p};
                const uint8_t* mask[2] = {sl.have_mask[0] ? d->lvl[l].mask[0].p : nullptr, sl.have_mask[1] ? d->lvl[l].mask[1].p : nullptr};
                if (st.njobs + 1 > kFeMaxJobs) flush_bits();
                if (top) {
                    const uint32_t bit0[2] = {lv.lm_off[0] - d->cbits_byte0, lv.lm_off[1] - d->cbits_byte0};
                    fe_job_top_bits(st.job[st.njobs++], quant, mask, d->cbits_arena[arena].p, bit0, B.W, B.H, lv.T, d->fe_top_mode);
                } else {
                    uint8_t* bits[2] = {d->bits_arena[arena].p + (lv.sm_off[0] >> 1), d->bits_arena[arena].p + (lv.sm_off[1] >> 1)};
                    fe_job_bits_rows(st.job[st.njobs++], quant, mask, bits, B.W, B.H, lv.T, d->fe_top_mode == 0);
                }
            }
        }
        flush_bits();
    }
    d->fe_bytes_low = !direct_low; d->fe_bytes_top = !direct_top;
    d->last_arena = first;                               // read_stage: the maps of level_bufs(0, .) belong to the batch's first frame
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

// The bit-plane refinement (match.hip): any pyramid with a level below the top; entries of up to 16383 features (two
// modalities of the reference's 8191, LL.cpp:1291).  LM_BITPLANES=0 / lm_detector_set_paths: the byte paths.
static bool bits_active(const lm_detector* d, int num_work) {
    return knobs().bitplanes && d->refine_mode == 0 && num_work > 0 && d->geom.levels >= 2 && d->bits_max_nf <= 16383;
}
// ... and the coarse pass on the pair stream of the top level (it plans no tiles, so only together with the bit-plane refinement)
static bool cbits_active(const lm_detector* d, int num_work) {
    return bits_active(d, num_work) && knobs().coarse_bits && d->coarse_mode == 0 && d->cbits_max_nf <= 16383;
}
// Device pointers of result slot `si` (everything a frame in flight owns).
static int frame_slot(lm_detector* d, int si, bool tiled, uint32_t tile_cap, FrameSlot* out) {
    lm_detector::Slot& sl = d->slot[si];
    const uint32_t cc = d->buf_cand_cap;
    FrameSlot F{};
    F.lm_arena = d->lm_arena[si].p; F.sm_arena = d->sm_arena[si].p;
    F.cands = d->d_cands.p + (size_t)cc * si;
    F.tiles = tiled ? d->d_tiles.p + (size_t)tile_cap * si : nullptr;
    F.todo = tiled ? d->d_todo.p + (size_t)cc * si : nullptr;
    F.counters = d->d_counters.p + (size_t)kCounterWords * si;
    F.matches_dev = d->d_matches_dev.p + (size_t)cc * si;
    F.dedupe_table = d->d_hash.p + dedupe_table_slots(cc) * (size_t)si;
    F.distinct_keys = d->d_distinct_keys.p + (size_t)cc * si;
    F.final_dev = d->d_final.p + 8 * (size_t)si;
    F.matches = nullptr;                                  // (k_local no longer stores the raw records into host memory: 16-byte PCIe writes per candidate)
    HIP_TRY(hipHostGetDevicePointer((void**)&F.distinct, sl.h_distinct, 0));
    HIP_TRY(hipHostGetDevicePointer((void**)&F.final_host, sl.h_counters, 0));
    *out = F;
    return LM_OK;
}

// Takes the next result slot for a frame (resident frame or ingest ring entry `ring`) and queues it behind the frames that wait for
// their batch; nothing is launched here.  The frames of a batch share threshold, work list and buffers, so a change of any of them
// launches what is waiting first.
static int slot_begin(lm_detector* d, float threshold, const char* const* class_ids, int num_class_ids, const uint8_t* rgb, const uint16_t* depth,
                      const bool have_mask[2], int ring) {
    if (d->n_submitted - d->n_collected >= (uint64_t)lm_detector::kSlots)
        return lm_set_error(LM_ERR_INVALID, "%d frames already in flight: call lm_detector_collect first", lm_detector::kSlots);
    HIP_TRY(hipSetDevice(d->device));
    int rc;
    if (d->bank_dirty || d->bank_geom_W != d->fW || d->bank_geom_H != d->fH) {
        if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "bank or frame geometry changed with a frame in flight");
        if ((rc = upload_bank(d))) return rc;
    }
    {   // the same selection as the frames waiting for their batch?  (build_work replaces the device-resident work list otherwise)
        std::vector<std::string> key;
        if (class_ids && num_class_ids > 0)
            for (int i = 0; i < num_class_ids; ++i) key.push_back(class_ids[i] ? class_ids[i] : "");
        const bool same = d->work_valid && key == d->work_key && d->work_key_rank == d->shard_rank && d->work_key_world == d->shard_world;
        if (d->pend_n && (!same || threshold != d->pend_threshold) && (rc = lm_launch_pending(d))) return rc;
    }
    if ((rc = build_work(d, class_ids, num_class_ids))) return rc;
    const int num_work = (int)d->work_pyr.size();
    const int K = lm_detector::kSlots;
    if (d->buf_cand_cap < d->cand_cap) {
        // first use, or the candidate capacity was raised after an overflow: the per-slot buffers are replaced.  Frames waiting for
        // their batch are launched, frames in flight finish on the old buffers first.
        if ((rc = lm_launch_pending(d))) return rc;
        if ((rc = sync_all_streams(d))) return rc;
        const uint32_t cc = d->cand_cap;
        if ((rc = d->d_cands.ensure((size_t)cc * K))) return rc;            // per result slot: coarse(k+1) runs beside local(k)
        if ((rc = d->d_matches_dev.ensure((size_t)cc * K))) return rc;
        if ((rc = d->d_hash.ensure(dedupe_table_slots(cc) * K))) return rc;   // one table per result slot
        if ((rc = d->d_distinct_keys.ensure((size_t)cc * K))) return rc;
        d->buf_cand_cap = cc;
    }
    if (!d->d_counters.p) {                                                          // per result slot; zero from here on (see k_dedupe)
        if ((rc = d->d_counters.ensure((size_t)kCounterWords * K))) return rc;
        if ((rc = d->d_final.ensure(8 * (size_t)K))) return rc;
        HIP_TRY(hipMemset(d->d_counters.p, 0, (size_t)kCounterWords * K * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(d->d_final.p, 0, 8 * (size_t)K * sizeof(unsigned long long)));
    }
    // tile refinement (match.hip): two-level pyramids with a tileable geometry; the buffers exist per result slot
    const bool tiled = (tiles_wanted(d) && num_work > 0 && tile_plan_possible(d->geom)) || bits_active(d, num_work);   // (the bit-plane path uses the todo bytes)
    const uint32_t tile_cap = d->buf_cand_cap / 2;      // a tile has at least two members
    if (tiled && (d->d_tiles.cap < (size_t)tile_cap * K || d->d_todo.cap < (size_t)d->buf_cand_cap * K)) {
        if ((rc = lm_launch_pending(d))) return rc;
        if ((rc = sync_all_streams(d))) return rc;
        if ((rc = d->d_tiles.ensure((size_t)tile_cap * K))) return rc;
        if ((rc = d->d_todo.ensure((size_t)d->buf_cand_cap * K))) return rc;
    }
    const int si = (int)(d->n_submitted % K);
    lm_detector::Slot& sl = d->slot[si];
    if ((rc = ensure_slot_buffers(d, sl, std::max<uint32_t>(sl.match_cap, d->buf_cand_cap)))) return rc;
    sl.t0 = std::chrono::steady_clock::now();
    sl.threshold = threshold; sl.num_work = num_work; sl.coarse_bytes = d->work_coarse_bytes; sl.h2d_ms = d->last_h2d_ms;
    sl.work_cls = d->work_cls; sl.work_tid = d->work_tid;
    sl.cand_cap = d->buf_cand_cap; sl.cands = d->d_cands.p + (size_t)d->buf_cand_cap * si;
    sl.matches_dev = d->d_matches_dev.p + (size_t)d->buf_cand_cap * si;
    sl.in_rgb = rgb; sl.in_depth = depth; sl.have_mask[0] = have_mask[0]; sl.have_mask[1] = have_mask[1]; sl.ring = ring;
    sl.launched = false; sl.pending = true; sl.leader = -1; sl.batch_n = 0;
    if (d->pend_n == 0) { d->pend_first = si; d->pend_threshold = threshold; }
    ++d->pend_n;
    ++d->n_submitted;
    return LM_OK;
}

static inline double host_seconds(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(t.time_since_epoch()).count(); }

// Batches launched and not yet finished on the GPU (an event query per finished batch, none in the steady state of a full queue).
static int batches_queued(lm_detector* d) {
    while (!d->queued.empty() && hipEventQuery(d->slot[d->queued.front().slot].done) == hipSuccess) d->queued.erase(d->queued.begin());
    (void)hipGetLastError();                                  // hipErrorNotReady is not an error
    return (int)d->queued.size();
}

// GPU time of a batch of n frames: measured, or scaled from the nearest measured size (a batch costs about four frames' worth of
// fixed latency + its frames), or 0 when nothing has been measured yet.
static float batch_ms_estimate(const lm_detector* d, int n) {
    if (d->batch_ms[n] > 0.f) return d->batch_ms[n];
    for (int k = 1; k <= kMaxBatch; ++k)
        for (int m : {n - k, n + k})
            if (m >= 1 && m <= kMaxBatch && d->batch_ms[m] > 0.f) return d->batch_ms[m] * (4.f + (float)n) / (4.f + (float)m);
    return 0.f;
}

// Should the frames waiting for their batch go out now?  `at` = host time the question is asked for.
//   * nothing launched is still uncollected, or everything launched has finished: the GPU is idle, the waiting frames go out (the
//     first frame of a stream, a caller that collects every frame before the next) — in a tight loop once LM_FIRST_BATCH (3) of them
//     wait, and once per burst: the launch occupies the caller for two submits' worth of time and a lone frame costs the GPU twice a
//     batched one (A/B at the driver's 20 steps: 0.1054 -> 0.0997 ms per frame, 200 steps unchanged; profiles/r04_stream_ab.txt);
//   * the caller submits in a tight loop (frames arrive less than 2.5 launches' worth of host time apart): only full batches.  A
//     launch costs the calling thread ~0.1 ms (seven kernel launches + events) whatever the batch size, so a stream of partial
//     batches makes the HOST the bottleneck at the pace of one launch per frame, the GPU keeps up with it, looks about to run dry
//     at every submit — and the stream stays there (measured: 0.213 instead of 0.155 ms per frame).  lm_detector_collect launches
//     what is left when it is about to block on the last launched batch, so nothing waits for frames that never come;
//   * frames arrive sparsely (a camera): the GPU-time model — launch when the GPU's estimated backlog is shorter than the slack.
static bool partial_batch_due(lm_detector* d, double at) {
    if (d->pend_n <= 0 || d->keep_queued <= 0) return false;
    const bool tight = d->submit_gap_ms > 0.f && d->submit_gap_ms < 2.5f * d->launch_cost_ms;   // (submit_gap_ms 0: no second submit yet — sparse until shown otherwise)
    const bool drained = d->n_launched == d->n_collected;
    if (drained) d->early_batch_used = false;                  // nothing in flight: a new burst
    if (tight) {
        // ONE early batch per burst: the GPU is idle (nothing launched is unfinished), LM_FIRST_BATCH frames wait.  Not again until the pipeline has
        // drained: with a host that needs longer for three submits + a launch than the GPU for three frames, every early batch would find the GPU idle
        // again and the stream would settle on three frames per launch (one run in three of a 20-step series did: 0.154 instead of 0.100 ms per frame).
        if (d->early_batch_used || d->pend_n < knobs().first_batch) return false;
        if (!drained && batches_queued(d) != 0) return false;
        d->early_batch_used = true;
        return true;
    }
    // sparse: nothing launched is unfinished (collected or not) -> the GPU is idle, the frames go out; else the GPU-time model
    if (drained || batches_queued(d) == 0) return true;
    if (batch_ms_estimate(d, d->pend_n) <= 0.f) return batches_queued(d) < d->keep_queued;
    return d->gpu_free_at - at <= 1e-3 * d->launch_slack_ms;
}

// Ordering between the detector's two queues.  Every kernel of a batch runs on `mstream`; `stream` carries what the synchronous entry points
// enqueue — a blocking upload, the device-to-device copy of lm_detector_select_frame, the clearing of new arenas, a training view.  A batch must see
// all of that: whatever is still pending on `stream` when the batch is enqueued comes first.  (Nothing pending there — the steady state of a stream
// of uploaded frames — needs no ordering: a query instead of a record, a cross-queue wait and the barrier packet the GPU would process for it.)
// The other direction — work on `stream` that touches what a batch in flight reads or writes (level buffers, arenas, the resident frame) — is not
// ordered by events: such entry points run only with nothing in flight (n_submitted == n_collected, checked where they start) or wait for the
// batch's front end (select_frame: resident_reader).  A new caller that writes those buffers on `stream` has to do the same.
static int order_after_default_stream(lm_detector* d, hipStream_t s) {
    if (s == d->stream || hipStreamQuery(d->stream) == hipSuccess) return LM_OK;
    (void)hipGetLastError();                                  // hipErrorNotReady is not an error
    HIP_TRY(hipEventRecord(d->ev[5], d->stream));
    HIP_TRY(hipStreamWaitEvent(s, d->ev[5], 0));
    return LM_OK;
}

// Enqueue the whole device pipeline of the frames waiting in slots [pend_first, pend_first + pend_n): ONE front end, coarse pass,
// refinement and duplicate removal for all of them (asynchronous).
int lm_launch_pending(lm_detector* d) {
    const int nb = d->pend_n, first = d->pend_first;
    if (nb <= 0) return LM_OK;
#ifdef LM_DIAG
    const auto tp_launch0 = std::chrono::steady_clock::now();
#endif
    HIP_TRY(hipSetDevice(d->device));
    d->pend_n = 0;
    lm_detector::Slot& lead = d->slot[first];
    const int num_work = lead.num_work;
    const float threshold = lead.threshold;
    const bool bits = bits_active(d, num_work);
    const bool tiled = !bits && tiles_wanted(d) && num_work > 0 && tile_plan_possible(d->geom);
    const uint32_t tile_cap = d->buf_cand_cap / 2;
    FrameBatch fb{};
    fb.nb = nb;
    int rc;
    for (int b = 0; b < nb; ++b)
        if ((rc = frame_slot(d, (first + b) % lm_detector::kSlots, tiled, tile_cap, &fb.f[b]))) return rc;
    hipStream_t ms = d->mstream, s = ms;                      // every kernel of a batch on the matching stream (one queue; the end of a stage is the start of the next)
    // the frames' uploads (copy stream) before the front end
    for (int b = nb - 1; b >= 0; --b) {                       // (the copy stream is one in-order queue: the upload of the batch's last streamed frame covers the earlier ones)
        const int ring = d->slot[(first + b) % lm_detector::kSlots].ring;
        if (ring < 0) continue;
        if (hipEventQuery(d->ingest.t1[ring]) != hipSuccess) { (void)hipGetLastError(); HIP_TRY(hipStreamWaitEvent(s, d->ingest.t1[ring], 0)); }   // (already there: nothing to wait for)
        break;
    }
    if ((rc = order_after_default_stream(d, s))) return rc;
    HIP_TRY(hipEventRecord(lead.ev[0], s));
    const bool cbits = cbits_active(d, num_work);
    // The front end writes the bit planes directly where nothing reads the byte planes: below the top when no candidate can leave its planes
    // (then k_local never runs behind k_local_bits), at the top level when the coarse pass runs on the pair stream.
    bool direct_low = knobs().fe_bits && d->fe_direct && bits && d->bits_all_in, direct_top = knobs().fe_bits && d->fe_direct && cbits;
    for (int l = 0; l + 1 < d->geom.levels; ++l) direct_low = direct_low && fe_bits_rows_possible(d->geom.lv[l].W, d->geom.lv[l].T);
    const LevelGeom& topl = d->geom.lv[d->geom.levels - 1];
    const uint32_t top_bit0[2] = {topl.lm_off[0] - d->cbits_byte0, topl.lm_off[1] - d->cbits_byte0};
    const bool top_ored = direct_top && fe_top_bits_kind(topl.W, topl.H, topl.T, top_bit0, d->fe_top_mode) == kFeTopBits;   // (else whole bytes / dwords are stored: nothing to clear)
    if (top_ored)                                        // the pair stream is OR-ed together: it has to be zero (k_local_bits leaves it so; k_pack_top and first use do not)
        for (int b = 0; b < nb; ++b) {
            const int si = (first + b) % lm_detector::kSlots;
            if (!d->cbits_clean[si]) HIP_TRY(hipMemsetAsync(d->cbits_arena[si].p, 0, (size_t)d->cbits_npairs * 8, s));
            d->cbits_clean[si] = false;                  // dirty from the front end on, until k_local_bits (top_clear) is enqueued behind it: an error return in between must not leave it marked clean
        }
#ifdef LM_DIAG
    static double lp_t[6]; static long lp_n;
    auto lp_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double lp0 = host_seconds(tp_launch0), lp1 = lp_now();
#endif
    if ((rc = run_frontend_batch(d, first, nb, s, direct_low, direct_top))) return rc;
#ifdef LM_DIAG
    const double lp2 = lp_now();
#endif
    BitsBatch bb{};
    FrameBatch fb_rest = fb;                                  // for k_local's per-candidate path on what k_local_bits leaves (todo = 1)
    if (bits) {
        for (int b = 0; b < nb; ++b) {
            const int si = (first + b) % lm_detector::kSlots;
            bb.strips[b] = d->sm_arena[si].p; bb.bits[b] = d->bits_arena[si].p;
            fb.f[b].todo = fb_rest.f[b].todo = d->d_todo.p + (size_t)d->buf_cand_cap * si;
            fb_rest.f[b].tiles = d->d_tiles.p + (size_t)tile_cap * si;   // non-null: "only the candidates marked todo"; no tile was planned
        }
        if (!direct_low)
            for (int l = 0; l + 1 < d->geom.levels; ++l) launch_pack_bits(bb, nb, d->geom.lv[l], s);
    }
    TopBits tb{};
    if (cbits) {
        for (int b = 0; b < nb; ++b) {
            const int si = (first + b) % lm_detector::kSlots;
            tb.lm[b] = d->lm_arena[si].p; tb.bits[b] = d->cbits_arena[si].p;
            if (top_ored && !d->fe_keep_top) bb.top_clear[b] = d->cbits_arena[si].p;     // zeroed again by k_local_bits, after k_coarse_bits has read it
            else d->cbits_clean[si] = false;
        }
        if (top_ored && !d->fe_keep_top) bb.top_clear_units = (d->cbits_npairs * 8u + 15u) / 16u;
        if (!direct_top) launch_pack_top(tb, nb, d->cbits_byte0, d->cbits_npairs, s);
    }
    // One queue for the whole batch: the end of a stage IS the start of the next — one timing record between two kernels instead of two or
    // three, and fe_done only when something outside the batch waits for this front end (a resident frame).
    bool resident_in = false;
    for (int b = 0; b < nb; ++b) resident_in = resident_in || d->slot[(first + b) % lm_detector::kSlots].ring < 0;
    HIP_TRY(hipEventRecord(lead.ev[1], s));
    if (resident_in) HIP_TRY(hipEventRecord(lead.fe_done, s));
    for (int b = 0; b < nb; ++b)                              // the resident frame is read by this front end: the next lm_detector_select_frame copy waits for it
        if (d->slot[(first + b) % lm_detector::kSlots].ring < 0) d->resident_reader = lead.fe_done;
    for (int b = 0; b < nb; ++b) {                            // a resident re-match of a streamed frame reads its ring entry: the entry's next upload waits for this front end
        const lm_detector::Slot& sl = d->slot[(first + b) % lm_detector::kSlots];
        if (sl.ring < 0 && d->ingest.stream)
            for (int r = 0; r < lm_detector::kSlots; ++r)
                if (d->ingest.d_rgb[r].p && sl.in_rgb == d->ingest.d_rgb[r].p) d->ingest.reader[r] = lead.fe_done;
    }
    const uint32_t cap = std::min<uint32_t>(lead.match_cap, d->buf_cand_cap);
    auto enqueue_coarse = [&](hipStream_t st) -> int {
        // the counters are zero on entry (reset by the slots' previous k_dedupe)
        { LM_CLOCK("launch_coarse");
        if (cbits) launch_coarse_bits(fb, tb, d->geom, d->d_entries.p, d->d_feat_off.p, d->d_work.p, num_work, threshold, d->buf_cand_cap, d->cbits_byte0, d->cbits_max_nf, st);
        else launch_coarse(fb, d->geom, d->d_entries.p, d->d_feat_off.p, d->d_work.p, num_work, threshold, d->buf_cand_cap, tile_cap, st);
        }
        HIP_TRY(hipEventRecord(lead.ev[3], st));
        return LM_OK;
    };
    auto enqueue_match = [&]() -> int {
        // persistent refinement grid over the tiles and then the remaining candidates of every frame of the batch; the counts are
        // read on the device (no host round trip), the records stored straight into the slots' pinned host memory; it also empties
        // the hash tables k_dedupe uses
        if (bits) {
            { LM_CLOCK("launch_local_bits");
            launch_local_bits(fb, bb, d->geom, d->d_entries.p, d->d_feat_word.p, d->d_work.p, d->buf_cand_cap, threshold, cap,
                              (uint32_t)dedupe_table_slots(d->buf_cand_cap), bits_grid(d, nb), d->bits_max_nf, ms); }
            if (bb.top_clear_units)       // the pair streams this launch zeroes again are clean for their slots' next frames
                for (int b = 0; b < nb; ++b)
                    if (bb.top_clear[b]) d->cbits_clean[(first + b) % lm_detector::kSlots] = true;
            if (!d->bits_all_in)          // candidates whose windows leave their planes (marked in todo): k_local's per-candidate path
                launch_local(fb_rest, d->geom, d->d_entries.p, d->d_feat_off.p, d->d_feat_word.p, d->d_run_mask.p, d->d_feat_xy.p, d->d_work.p, d->buf_cand_cap, threshold, cap,
                             (uint32_t)dedupe_table_slots(d->buf_cand_cap), tile_cap, d->num_cus * 2, ms);
        } else
        if (num_work > 0)
            launch_local(fb, d->geom, d->d_entries.p, d->d_feat_off.p, d->d_feat_word.p, d->d_run_mask.p, d->d_feat_xy.p, d->d_work.p, d->buf_cand_cap, threshold, cap,
                         (uint32_t)dedupe_table_slots(d->buf_cand_cap), tile_cap, local_grid(d, nb), ms);
        HIP_TRY(hipEventRecord(lead.ev[4], ms));
        return LM_OK;
    };
    // exact duplicates out (they never survive std::unique): distinct records + counts to the slots' pinned memory
    // k_dedupe's grid per frame: a workgroup per 256 candidates of the LAST collected frame (the kernel strides over whatever the count turns out to
    // be), between 64 and two per CU.  Every workgroup takes a ticket at the frame's counter and most of a 2-per-CU grid had nothing else to do:
    // 31 -> 19.5 us per 8-frame batch at 16k candidates per frame (profiles/r04_stream_ab.txt).
    const int dedupe_blocks = knobs().dedupe_blocks > 0 ? knobs().dedupe_blocks
                                                        : std::max(64, std::min(d->num_cus * 2, (int)((d->ncand_hint + 255) / 256)));
    auto enqueue_dedupe = [&](hipStream_t st) -> int {
        if (num_work > 0) {
            LM_CLOCK("launch_dedupe");
            launch_dedupe(fb, d->buf_cand_cap, dedupe_table_slots(d->buf_cand_cap), d->d_work_cls.p, d->d_work_tid.p, dedupe_blocks, st);
        }
        else
            for (int b = 0; b < nb; ++b) HIP_TRY(hipMemsetAsync(fb.f[b].final_dev, 0, 8 * sizeof(unsigned long long), st));   // nothing searched: no records for NMS / exchange
        return LM_OK;
    };
    {
#ifdef LM_DIAG
        const double lp3 = lp_now();
#endif
        if ((rc = enqueue_coarse(ms))) return rc;
#ifdef LM_DIAG
        const double lp4 = lp_now();
#endif
        if ((rc = enqueue_match())) return rc;
#ifdef LM_DIAG
        const double lp5 = lp_now();
#endif
        if ((rc = enqueue_dedupe(ms))) return rc;
        { LM_CLOCK("record done"); HIP_TRY(hipEventRecord(lead.done, ms)); }
#ifdef LM_DIAG
        const double lp6 = lp_now();
        if (getenv("LM_LAUNCH_SERIES") && lp_n < 80) fprintf(stderr, "batch %ld nb %d: %.0f us (fe %.0f coarse %.0f refine %.0f dedupe %.0f)\n", lp_n, nb, 1e6 * (lp6 - lp0), 1e6 * (lp2 - lp1), 1e6 * (lp4 - lp3), 1e6 * (lp5 - lp4), 1e6 * (lp6 - lp5));
        lp_t[0] += lp1 - lp0; lp_t[1] += lp2 - lp1; lp_t[2] += lp3 - lp2; lp_t[3] += lp4 - lp3; lp_t[4] += lp5 - lp4; lp_t[5] += lp6 - lp5; ++lp_n;
        if (getenv("LM_LAUNCH_PROF") && lp_n % 32 == 0)
            fprintf(stderr, "launch profile over %ld batches (us): waits+ev0 %.1f | front end %.1f | bits tables+ev1 %.1f | coarse %.1f | refine %.1f | dedupe+done %.1f\n", lp_n,
                    1e6 * lp_t[0] / lp_n, 1e6 * lp_t[1] / lp_n, 1e6 * lp_t[2] / lp_n, 1e6 * lp_t[3] / lp_n, 1e6 * lp_t[4] / lp_n, 1e6 * lp_t[5] / lp_n);
#endif
    }
    const auto now = std::chrono::steady_clock::now();
    for (int b = 0; b < nb; ++b) {
        lm_detector::Slot& sl = d->slot[(first + b) % lm_detector::kSlots];
        sl.launched = true; sl.leader = first; sl.batch_n = nb; sl.t1 = now;
    }
    {
        const double t = host_seconds(now);
        const bool idle = batches_queued(d) == 0;
        if (idle) d->gpu_free_at = std::min(d->gpu_free_at, t);
        d->gpu_free_at = std::max(d->gpu_free_at, t) + 1e-3 * batch_ms_estimate(d, nb);
        d->queued.push_back({d->n_launched, first, nb, t, idle});
    }
    d->n_launched += (uint64_t)nb;
    return LM_OK;
}

// The detector's current frame (lm_detector_set_frame / select_frame, or the frame a previous submit_frame left current) as a
// batch of one, launched at once.
int lm_submit_frame(lm_detector* d, float threshold, const char* const* class_ids, int num_class_ids) {
    if (!d->frame_valid) return lm_set_error(LM_ERR_INVALID, "no frame resident: call lm_detector_set_frame / select_frame first");
    int rc;
    if ((rc = lm_launch_pending(d))) return rc;               // frames waiting for their batch go first (results come back in order)
    if ((rc = slot_begin(d, threshold, class_ids, num_class_ids, d->cur_rgb, d->cur_depth, d->have_mask, -1))) return rc;
    return lm_launch_pending(d);
}

// The distinct records of a finished frame (k_dedupe's output in the slot's pinned memory) as lm_match, canonically sorted + uniqued
// (Detector::match's list under the canonical order of SURVEY A12).  Returns the count; *res is malloc'ed.
static size_t canonical_list_of(const lm_detector::Slot& sl, uint64_t nd, lm_match** res_out, float* convert_ms, float* merge_ms) {
    const auto t0 = std::chrono::steady_clock::now();
    lm_match* res = (lm_match*)malloc(std::max<size_t>(1, (size_t)nd) * sizeof(lm_match));
    *res_out = res;
    if (!res) return 0;
    const std::vector<int32_t>& wcls = *sl.work_cls;
    const std::vector<int32_t>& wtid = *sl.work_tid;
    size_t w = 0;
    const Candidate* src = sl.h_distinct;
    for (uint64_t i = 0; i < nd; ++i) {
        const Candidate& c = src[i];
        if (c.work < 0) continue;
        res[w].x = c.x; res[w].y = c.y; res[w].similarity = c.score;
        res[w].class_index = wcls[c.work];
        res[w].template_id = wtid[c.work];
        ++w;
    }
    const auto t1 = std::chrono::steady_clock::now();
    const size_t n = merge_matches_impl(res, w, true);
    const auto t2 = std::chrono::steady_clock::now();
    if (convert_ms) *convert_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    if (merge_ms) *merge_ms = std::chrono::duration<float, std::milli>(t2 - t1).count();
    return n;
}

// ---- helper threads of the streamed path (detector_internal.h, HostPool): no HIP calls on them ----
static void pool_main(lm_detector* d) {
    lm_detector::HostPool& P = d->pool;
    for (;;) {
        std::function<void()> job;
        int spins = 0;
        for (;;) {
            if (P.posted.load(std::memory_order_acquire) > 0) {
                std::lock_guard<std::mutex> lk(P.mu);
                if (!P.jobs.empty()) { job = std::move(P.jobs.front()); P.jobs.pop_front(); P.posted.fetch_sub(1, std::memory_order_acq_rel); break; }
            }
            if (++spins < 30000) { __builtin_ia32_pause(); continue; }          // ~0.3 ms of spinning, then sleep
            std::unique_lock<std::mutex> lk(P.mu);
            if (P.stop) return;
            P.asleep.fetch_add(1, std::memory_order_seq_cst);
            P.cv.wait(lk, [&] { return P.stop || !P.jobs.empty(); });
            P.asleep.fetch_sub(1, std::memory_order_seq_cst);
            if (P.stop && P.jobs.empty()) return;
            spins = 0;
        }
        job();
    }
}
// Starts the helpers on first use.  Never more than the CPUs this process may run on leave free (a cgroup / affinity mask of a few cores, eight
// ranks on one node), and a thread the system refuses (std::system_error: a container's thread limit) only shrinks the pool: the streamed path
// works without helpers (the caller copies and sorts on its own).
static bool pool_ready(lm_detector* d) {
    lm_detector::HostPool& P = d->pool;
    if (P.threads <= 0) return false;
    if (!P.started) {
        int cpus = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        P.threads = std::max(0, std::min(P.threads, cpus - 1));        // one CPU stays with the calling thread
        P.stop = false;
        int started = 0;
        for (int i = 0; i < P.threads; ++i) {
            try { P.th.emplace_back(pool_main, d); ++started; }
            catch (const std::system_error&) { break; }
        }
        P.threads = started;
        P.started = started > 0;
    }
    return P.threads > 0;
}
// The calling thread takes a queued job itself (while it waits for the helpers: a helper that was descheduled must not hold the caller up)
static bool pool_run_one(lm_detector* d) {
    lm_detector::HostPool& P = d->pool;
    if (P.posted.load(std::memory_order_acquire) <= 0) return false;
    std::function<void()> job;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (P.jobs.empty()) return false;
        job = std::move(P.jobs.front()); P.jobs.pop_front(); P.posted.fetch_sub(1, std::memory_order_acq_rel);
    }
    job();
    return true;
}
static void pool_post(lm_detector* d, std::function<void()> job) {
    lm_detector::HostPool& P = d->pool;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.jobs.push_back(std::move(job));
        P.posted.fetch_add(1, std::memory_order_seq_cst);
    }
    if (P.asleep.load(std::memory_order_seq_cst) > 0) P.cv.notify_one();
}
static void pool_stop(lm_detector* d) {
    lm_detector::HostPool& P = d->pool;
    if (!P.started) return;
    { std::lock_guard<std::mutex> lk(P.mu); P.stop = true; }
    P.cv.notify_all();
    for (auto& t : P.th) if (t.joinable()) t.join();
    P.th.clear();
    P.started = false;
}

// memcpy into a pinned staging buffer with non-temporal stores: the buffer is read next by the copy engine, not by a core, and a slice
// (a few hundred KB) is below the size from which glibc's memcpy streams on its own — ordinary stores first READ every destination line
// (read for ownership).  Falls back to memcpy for small or unaligned pieces and on hosts without AVX2.
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
__attribute__((target("avx2"))) static void copy_stream_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64)), e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a); _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c); _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), e);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}
static void copy_staging(uint8_t* dst, const uint8_t* src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && knobs().nt_copy;
    if (avx2 && n >= 16384 && (reinterpret_cast<uintptr_t>(dst) & 31) == 0) copy_stream_avx2(dst, src, n);
    else memcpy(dst, src, n);
}
#else
static void copy_staging(uint8_t* dst, const uint8_t* src, size_t n) { memcpy(dst, src, n); }
#endif

// dst <- a, dst_b <- b (the two images of a frame), cut into slices for the caller's thread and the helpers
static void staged_copy(lm_detector* d, uint8_t* dst, const uint8_t* a, size_t na, uint8_t* dst_b, const uint8_t* b, size_t nb) {
    const bool same_a = a == dst, same_b = b == dst_b;              // zero-copy: the caller filled lm_detector_ingest_buffer's pointers
    if (same_a || same_b || na + nb < (1u << 19) || !pool_ready(d)) {   // small frames: one thread
        if (!same_a) memcpy(dst, a, na);
        if (!same_b) memcpy(dst_b, b, nb);
        return;
    }
    const int parts = d->pool.threads + 1;
    const size_t total = na + nb, per = ((total + (size_t)parts - 1) / (size_t)parts + 4095) & ~(size_t)4095;
    auto copy_range = [=](size_t lo, size_t hi) {                   // bytes [lo, hi) of the two images taken as one run
        if (lo < na) copy_staging(dst + lo, a + lo, std::min(hi, na) - lo);
        if (hi > na) { const size_t l2 = std::max(lo, na); copy_staging(dst_b + (l2 - na), b + (l2 - na), hi - l2); }
    };
    std::atomic<int> left{0};
    int posted = 0;
    for (int p = 1; p < parts; ++p) {
        const size_t lo = std::min(total, per * (size_t)p), hi = std::min(total, per * (size_t)(p + 1));
        if (lo >= hi) break;
        left.fetch_add(1, std::memory_order_relaxed);
        ++posted;
        std::atomic<int>* lp = &left;
        pool_post(d, [=]() { copy_range(lo, hi); lp->fetch_sub(1, std::memory_order_release); });
    }
    copy_range(0, std::min(total, per));
    // queued jobs nobody has taken yet are run here — the copy slices above, and whatever else is queued: a list-preparation job of another
    // slot (~35 us, no HIP calls) may so run inside this submit; results do not depend on who runs a job —; then a bounded spin for the
    // slices in progress, then the CPU is given up between looks
    for (int spin = 0; left.load(std::memory_order_acquire) != 0;) {
        if (pool_run_one(d)) continue;
        if (++spin < 4000) __builtin_ia32_pause(); else std::this_thread::yield();
    }
    (void)posted;
}

// The canonical list of a finished frame on a helper thread (its records are in pinned memory: the caller has seen the batch's event)
static void prepare_list_job(lm_detector::Slot* sl) {
    int state = 2;
    sl->prep = nullptr; sl->prep_n = 0;
    const unsigned long long* hc = sl->h_counters;
    if (sl->num_work > 0 && hc[0] <= sl->cand_cap && hc[0] <= sl->match_cap && hc[1] <= hc[0]) {
        sl->prep_n = canonical_list_of(*sl, hc[1], &sl->prep, &sl->prep_collect_ms, &sl->prep_merge_ms);
        if (sl->prep) state = 1;
    }
    sl->ready.store(state, std::memory_order_release);
}

// Wait for the oldest frame in flight and turn its records into lm_match.  Returns 1 when a buffer
// overflowed (capacity has been raised; the frame has to be submitted again), 0 on success.
int lm_collect_frame(lm_detector* d, int sort_unique, lm_match** out, size_t* n_out) {
    if (d->n_collected == d->n_submitted) return lm_set_error(LM_ERR_INVALID, "no frame in flight");
    const auto t_enter = std::chrono::steady_clock::now();
    const int slot_index = (int)(d->n_collected % lm_detector::kSlots);
    lm_detector::Slot& sl = d->slot[slot_index];
    HIP_TRY(hipSetDevice(d->device));
    if (!sl.launched) {                                 // still waiting for its batch to fill: launch what is there
        int rc = lm_launch_pending(d);
        if (rc) return rc;
    }
    lm_detector::Slot& lead = d->slot[sl.leader];       // the events are those of the batch's first slot
    lm_match* prepared = nullptr;
    size_t prepared_n = 0;
    bool have_prepared = false;
    if (sl.prep_queued) {                               // a helper thread owns the slot until it has marked it ready (a job of ~35 us, posted when the batch's first frame was collected)
        for (int spin = 0; sl.ready.load(std::memory_order_acquire) == 0;) {    // (a job still queued — this slot's, possibly — is run here)
            if (pool_run_one(d)) continue;
            if (++spin < 20000) __builtin_ia32_pause(); else std::this_thread::yield();
        }
        have_prepared = sl.ready.load(std::memory_order_acquire) == 1;
        prepared = sl.prep; prepared_n = sl.prep_n;
        sl.prep = nullptr; sl.prep_n = 0; sl.prep_queued = false;
    }
    // Keep the GPU-time model current.  If the wait below blocks on the first frame of a batch, the batch finished when the wait
    // returned: that pins the estimate of when the GPU runs dry and — with the start of the batch known too (the previous batch's
    // end seen the same way, or an idle GPU at launch) — gives the batch's duration.  Frames waiting for their batch go out BEFORE
    // the wait if the GPU would have (almost) nothing left when it ends, or after it if it has by then.
    const bool batch_head = !d->queued.empty() && d->queued.front().first_frame == d->n_collected && d->queued.front().slot == sl.leader;
    bool blocked = false;
    lm_detector::QueuedBatch head{};
    auto later_ms = [&]() {                              // estimated GPU time of the batches launched after this frame's
        double ms = 0.0;
        for (const lm_detector::QueuedBatch& q : d->queued)
            if (q.first_frame > d->n_collected) { const float e = batch_ms_estimate(d, q.frames); ms += e > 0.f ? e : 1e3; }   // not timed yet: plenty
        return ms;
    };
    if (batch_head) {
        head = d->queued.front();
        blocked = hipEventQuery(lead.done) == hipErrorNotReady;
        (void)hipGetLastError();
        if (blocked && d->pend_n > 0 && d->keep_queued > 0 && later_ms() <= d->launch_slack_ms) {   // (no batch launched after this one: 0, with or without a GPU-time model)
            int rc = lm_launch_pending(d);
            if (rc) return rc;
        }
    }
    // (blocking wait: polling the event with hipEventQuery instead was slower, 0.107 against 0.092 ms per frame — profiles/r04_stream_ab.txt)
    HIP_TRY(hipEventSynchronize(lead.done));
    const auto t2 = std::chrono::steady_clock::now();
    // The batch has finished: the records of ALL its frames are in pinned memory.  The helper threads prepare the lists of the later frames
    // while this thread does this frame's (sort_unique = 1, the Detector.match list, is what a stream asks for frame after frame).
    if (sort_unique == 1 && d->async_collect && !d->reference_order && sl.leader == slot_index && sl.batch_n > 1 && sl.num_work > 0 && pool_ready(d))
        for (int b = 1; b < sl.batch_n; ++b) {
            lm_detector::Slot& later = d->slot[(slot_index + b) % lm_detector::kSlots];
            if (!later.pending || !later.launched || later.leader != slot_index || later.prep_queued) continue;
            later.ready.store(0, std::memory_order_relaxed);
            later.prep_queued = true;
            lm_detector::Slot* lp = &later;
            pool_post(d, [lp]() { prepare_list_job(lp); });
        }
    {
        const double now = host_seconds(t2), dry_at = now + 1e-3 * later_ms();
        if (batch_head && blocked) {
            const bool start_known = head.gpu_idle_at_launch || (d->last_done_at >= 0.0 && d->last_done_end == head.first_frame);
            if (start_known) {
                const double start = head.gpu_idle_at_launch ? head.launched_at : std::max(d->last_done_at, head.launched_at);
                const float ms = (float)((now - start) * 1e3);
                float& e = d->batch_ms[head.frames];
                if (ms > 0.f && ms < 1e3f) e = e > 0.f ? 0.75f * e + 0.25f * ms : ms;
            }
            d->gpu_free_at = dry_at;
            d->last_done_at = now;
            d->last_done_end = head.first_frame + (uint64_t)head.frames;
        } else {
            d->gpu_free_at = std::min(d->gpu_free_at, dry_at);
            if (batch_head) { d->last_done_at = -1.0; d->last_done_end = head.first_frame + (uint64_t)head.frames; }
        }
        const bool idle_after = d->n_launched == d->n_collected + 1;         // this was the last launched frame: the GPU has nothing left
        if (d->pend_n > 0 && d->keep_queued > 0 && (idle_after || (batch_ms_estimate(d, d->pend_n) > 0.f && partial_batch_due(d, now)))) {
            int rc = lm_launch_pending(d);
            if (rc) return rc;
        }
    }
    sl.pending = false;
    while (!d->queued.empty() && d->queued.front().first_frame <= d->n_collected) d->queued.erase(d->queued.begin());   // this frame's batch and everything before it are done
    if (d->xchg.state[slot_index] != 0) {               // exchange work of this frame may still read the slot's buffers
        HIP_TRY(hipStreamSynchronize(d->xchg.stream));
        d->xchg.state[slot_index] = 0;
    }
    ++d->n_collected;
    HIP_TRY(hipGetLastError());
    // published by the last block of the frame's k_dedupe: candidates, distinct, alive, key overflow, tiles, evaluations, bytes
    const unsigned long long* hc = sl.h_counters;
    const uint64_t ncand = sl.num_work > 0 ? hc[0] : 0;
    if (ncand > 0xFFFFFFF0ull) return lm_set_error(LM_ERR_INVALID, "too many coarse candidates (%llu)", (unsigned long long)ncand);
    if (ncand > sl.cand_cap || ncand > sl.match_cap) {   // never drop silently: grow, caller reruns the frame
        d->cand_cap = std::max<uint32_t>(d->cand_cap, (uint32_t)(ncand + ncand / 4 + 1024));
        d->ingest.used[slot_index] = false;
        free(prepared);
        return 1;
    }
    lm_timings tm{};
    tm.h2d_ms = sl.h2d_ms; tm.templates = sl.num_work; tm.coarse_bytes = sl.coarse_bytes;
    if (d->ingest.used[slot_index]) {   // streamed frame: its H2D ran on the copy stream
        d->ingest.used[slot_index] = false;
        float h = 0.f;
        if (hipEventElapsedTime(&h, d->ingest.t0[slot_index], d->ingest.t1[slot_index]) == hipSuccess) tm.h2d_ms = h;
    }
    const uint64_t evals = sl.num_work > 0 ? hc[5] : 0, lbytes = sl.num_work > 0 ? hc[6] : 0;
    uint64_t nm = 0;
    const Candidate* hm = sl.h_matches;
    if (sl.num_work > 0 && ncand > 0 && (sort_unique == 0 || sort_unique == 3 || d->reference_order))   // the raw per-candidate records, for the callers that want them
        HIP_TRY(hipMemcpy(sl.h_matches, sl.matches_dev, (size_t)ncand * sizeof(Candidate), hipMemcpyDeviceToHost));
    if (sl.num_work == 0) nm = 0;
    else if (sort_unique == 0) { for (uint64_t i = 0; i < ncand; ++i) nm += hm[i].work >= 0; }
    else nm = hc[2];                                   // counted on the device by k_dedupe: no pass over the raw records
    tm.coarse_candidates = (int64_t)ncand;
    d->ncand_hint = (uint64_t)ncand;
    tm.local_evals = (int64_t)evals;
    tm.local_bytes = (int64_t)lbytes;
    tm.matches_pre_unique = (int64_t)nm;
    tm.d2h_ms = 0.f;                                   // the records are stored straight into pinned memory by the refinement
    tm.batch_frames = sl.batch_n;
    if (hipEventElapsedTime(&tm.frontend_ms, lead.ev[0], lead.ev[1]) != hipSuccess ||
        hipEventElapsedTime(&tm.coarse_ms, lead.ev[1], lead.ev[3]) != hipSuccess ||
        hipEventElapsedTime(&tm.local_ms, lead.ev[3], lead.ev[4]) != hipSuccess ||
        hipEventElapsedTime(&tm.total_ms, lead.ev[0], lead.ev[4]) != hipSuccess) {
        (void)hipGetLastError();
        tm.frontend_ms = tm.coarse_ms = tm.local_ms = tm.d2h_ms = tm.total_ms = 0.f;
    }
    // the stage times are those of the LAUNCHES, which serve batch_frames frames: per frame = time / batch_frames
    if (sort_unique < 0) {                            // pipeline mode: the records stay on the device
        free(prepared);
        d->timings = tm;
        if (out) *out = nullptr;
        if (n_out) *n_out = 0;
        return LM_OK;
    }
    if (have_prepared && sort_unique == 1 && !d->reference_order) {   // the collector thread has the list ready: hand it over
        tm.host_submit_ms = std::chrono::duration<float, std::milli>(sl.t1 - sl.t0).count();
        tm.host_wait_ms = std::chrono::duration<float, std::milli>(t2 - sl.t1).count();
        tm.host_collect_ms = sl.prep_collect_ms; tm.host_merge_ms = sl.prep_merge_ms;   // spent on the collector thread
        d->timings = tm;
        *out = prepared; *n_out = prepared_n;
        return LM_OK;
    }
    free(prepared);
    // sort_unique = 0: every record alive (the raw pre-unique multiset); 1 / 2: the records without exact duplicates
    // (k_dedupe) — what std::unique would leave of them anyway — canonically sorted + uniqued (1) or as they are (2);
    // 3: the reference's own output, permutation and surviving duplicates included (below)
    if (sort_unique == 1 && d->reference_order) sort_unique = 3;
    if (sort_unique == 3) {
        // Detector::match ends with std::sort under an order that ignores x, y and std::unique under an equality that ignores
        // template_id (LL.cpp:1771-1776, LL.h:234-246): what comes out depends on the order the records went in and on
        // libstdc++'s introsort.  Both are reproducible: the reference appends class by class (caller's order), template by
        // template, candidates in raster order of the coarse grid (LL.cpp:1753-1769, 1835-1852; remove_if keeps the order) — the
        // coarse position of every slot is in the candidate buffer — and std::sort is the same template of the same libstdc++
        // this library is built with, so the same comparisons on the same sequence give the same permutation.
        std::vector<Candidate> coarse((size_t)ncand);                            // from the buffer the frame was submitted with: d->cand_cap may have grown since
        if (ncand) HIP_TRY(hipMemcpy(coarse.data(), sl.cands, (size_t)ncand * sizeof(Candidate), hipMemcpyDeviceToHost));
        const std::vector<int32_t>& wcls = *sl.work_cls;
        const std::vector<int32_t>& wtid = *sl.work_tid;
        struct Rec { int32_t cls, tid, cy, cx; lm_match m; };
        std::vector<Rec> recs;
        recs.reserve((size_t)nm);
        for (uint64_t i = 0; i < ncand; ++i) {
            const Candidate& c = hm[i];
            if (c.work < 0) continue;
            Rec r;
            r.cls = wcls[c.work]; r.tid = wtid[c.work]; r.cy = coarse[i].y; r.cx = coarse[i].x;
            r.m.x = c.x; r.m.y = c.y; r.m.similarity = c.score; r.m.class_index = r.cls; r.m.template_id = r.tid;
            recs.push_back(r);
        }
        std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {      // a total order: emission order of the reference
            if (a.cls != b.cls) return a.cls < b.cls;
            if (a.tid != b.tid) return a.tid < b.tid;
            if (a.cy != b.cy) return a.cy < b.cy;
            return a.cx < b.cx;
        });
        const auto t3 = std::chrono::steady_clock::now();
        std::vector<lm_match> v(recs.size());
        for (size_t i = 0; i < recs.size(); ++i) v[i] = recs[i].m;
        std::sort(v.begin(), v.end(), [](const lm_match& a, const lm_match& b) {  // Match::operator< (LL.h:234-241)
            if (a.similarity != b.similarity) return a.similarity > b.similarity;
            return a.template_id < b.template_id;
        });
        v.erase(std::unique(v.begin(), v.end(), match_eq), v.end());               // Match::operator== (LL.h:243-246)
        const auto t4 = std::chrono::steady_clock::now();
        lm_match* res = (lm_match*)malloc(std::max<size_t>(1, v.size()) * sizeof(lm_match));
        if (!res) return lm_set_error(LM_ERR_INVALID, "out of host memory");
        if (!v.empty()) memcpy(res, v.data(), v.size() * sizeof(lm_match));
        auto msf = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<float, std::milli>(b - a).count();
        };
        tm.host_submit_ms = msf(sl.t0, sl.t1); tm.host_wait_ms = msf(sl.t1, t2); tm.host_collect_ms = msf(t2, t3); tm.host_merge_ms = msf(t3, t4);
        d->timings = tm;
        *out = res; *n_out = v.size();
        return LM_OK;
    }
    const bool use_distinct = sort_unique != 0 && sl.num_work > 0;
    const uint64_t nd = use_distinct ? hc[1] : 0;
    if (use_distinct && (nd > ncand || nd > nm || nm > ncand))
        return lm_set_error(LM_ERR_HIP, "duplicate removal out of step with the refinement (%llu distinct of %llu alive, %llu candidates)",
                            (unsigned long long)nd, (unsigned long long)nm, (unsigned long long)ncand);
    const size_t nrec = use_distinct ? (size_t)nd : (size_t)nm;
    lm_match* res = (lm_match*)malloc(std::max<size_t>(1, nrec) * sizeof(lm_match));
    if (!res) return lm_set_error(LM_ERR_INVALID, "out of host memory");
    const std::vector<int32_t>& wcls = *sl.work_cls;
    const std::vector<int32_t>& wtid = *sl.work_tid;
    size_t w = 0;
    const Candidate* src = use_distinct ? sl.h_distinct : hm;
    const uint64_t nsrc = use_distinct ? nd : ncand;
    for (uint64_t i = 0; i < nsrc; ++i) {
        const Candidate& c = src[i];
        if (c.work < 0) continue;                     // dropped below the threshold during refinement
        res[w].x = c.x; res[w].y = c.y; res[w].similarity = c.score;
        res[w].class_index = wcls[c.work];
        res[w].template_id = wtid[c.work];
        ++w;
    }
    size_t n = w;
    const auto t3 = std::chrono::steady_clock::now();
    if (sort_unique == 1) n = merge_matches_impl(res, w, use_distinct);
    const auto t4 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<float, std::milli>(b - a).count();
    };
    tm.host_submit_ms = ms(sl.t0, sl.t1);
    tm.host_wait_ms = ms(sl.t1, t2);       // includes whatever the caller did between submit and collect
    tm.host_collect_ms = ms(t2, t3);
    tm.host_merge_ms = ms(t3, t4);
    d->host_prof[5] += std::chrono::duration<double>(t2 - t_enter).count();
    d->host_prof[6] += tm.host_collect_ms * 1e-3; d->host_prof[7] += tm.host_merge_ms * 1e-3;
    d->timings = tm;
    *out = res; *n_out = n;
    return LM_OK;
}

extern "C" int lm_detector_max_in_flight(void) { return lm_detector::kSlots; }

extern "C" int lm_detector_set_reference_order(lm_detector* d, int on) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    d->reference_order = on != 0;
    return LM_OK;
}

extern "C" int lm_detector_submit(lm_detector* d, float threshold, const char* const* class_ids, int num_class_ids) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    return lm_submit_frame(d, threshold, class_ids, num_class_ids);
}

// ---- live-stream ingest ---------------------------------------------------------------------------
// The per-frame call of a camera / dataset loop (linemod_ros/detect.py:83-138, linemod_and_levelup_test.py:314-327 hand a NEW
// host frame to every match): stage -> H2D on the copy stream -> front end + matching of lm_detector_submit, up to kSlots
// frames in flight; results come back through lm_detector_collect in submission order.
static int ingest_entry(lm_detector* d, int r, size_t n) {
    lm_detector::Ingest& g = d->ingest;
    if (!g.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        for (int i = 0; i < lm_detector::kSlots; ++i) { HIP_TRY(hipEventCreate(&g.t0[i])); HIP_TRY(hipEventCreate(&g.t1[i])); }
    }
    g.depth_off = (n * 3 + 15) & ~(size_t)15;                       // the depth image behind the colour image, 16-byte aligned (host entry and device entry alike)
    const size_t bytes = g.depth_off + n * 2;
    if (g.pinned_bytes[r] < bytes) {
        if (g.pinned[r]) (void)hipHostFree(g.pinned[r]);
        g.pinned[r] = nullptr; g.pinned_bytes[r] = 0;
        HIP_TRY(hipHostMalloc(&g.pinned[r], bytes, hipHostMallocDefault));
        g.pinned_bytes[r] = bytes;
    }
    int rc;
    if ((rc = g.d_rgb[r].ensure(bytes))) return rc;
    g.d_depth[r] = reinterpret_cast<uint16_t*>(g.d_rgb[r].p + g.depth_off);
    return LM_OK;
}

static int ingest_geometry(lm_detector* d, int width, int height) {
    if (width < 16 || height < 16 || width > 16384 || height > 16384) return lm_set_error(LM_ERR_INVALID, "unsupported frame size %dx%d", width, height);
    if (width != d->fW || height != d->fH || d->lm_arena[0].cap == 0) {
        if (d->n_submitted != d->n_collected)
            return lm_set_error(LM_ERR_INVALID, "frame size changes (%dx%d -> %dx%d) with frames in flight: collect them first", d->fW, d->fH, width, height);
        d->frame_valid = false;
        int rc = setup_geometry(d, width, height, true);
        if (rc) return rc;
    }
    return LM_OK;
}

extern "C" int lm_detector_ingest_buffer(lm_detector* d, int width, int height, uint8_t** rgb, uint16_t** depth) {
    if (!d || !rgb || !depth) return lm_set_error(LM_ERR_INVALID, "null argument");
    *rgb = nullptr; *depth = nullptr;
    if (d->n_submitted - d->n_collected >= (uint64_t)lm_detector::kSlots)
        return lm_set_error(LM_ERR_INVALID, "%d frames already in flight: call lm_detector_collect first", lm_detector::kSlots);
    HIP_TRY(hipSetDevice(d->device));
    int rc = ingest_geometry(d, width, height);
    if (rc) return rc;
    const int r = (int)(d->n_submitted % lm_detector::kSlots);
    const size_t n = (size_t)width * height;
    if ((rc = ingest_entry(d, r, n))) return rc;
    *rgb = (uint8_t*)d->ingest.pinned[r];
    *depth = (uint16_t*)((uint8_t*)d->ingest.pinned[r] + d->ingest.depth_off);
    return LM_OK;
}

extern "C" int lm_detector_submit_frame(lm_detector* d, const uint8_t* rgb, const uint16_t* depth, int width, int height, float threshold,
                                        const char* const* class_ids, int num_class_ids) {
    if (!d || !rgb || !depth) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (d->n_submitted - d->n_collected >= (uint64_t)lm_detector::kSlots)
        return lm_set_error(LM_ERR_INVALID, "%d frames already in flight: call lm_detector_collect first", lm_detector::kSlots);
    HIP_TRY(hipSetDevice(d->device));
    int rc = ingest_geometry(d, width, height);
    if (rc) return rc;
    const int r = (int)(d->n_submitted % lm_detector::kSlots);   // ring entry == result slot: free, its previous frame was collected
    const size_t n = (size_t)width * height;
    if ((rc = ingest_entry(d, r, n))) return rc;
    lm_detector::Ingest& g = d->ingest;
    uint8_t* st = (uint8_t*)g.pinned[r];
    const auto tp0 = std::chrono::steady_clock::now();
    staged_copy(d, st, rgb, n * 3, st + g.depth_off, (const uint8_t*)depth, n * 2);  // zero-copy when the caller filled lm_detector_ingest_buffer's pointers
    const auto tp1 = std::chrono::steady_clock::now();
    if (g.reader[r]) {                                            // a resident re-match of the entry's previous frame may still read it (another slot's front end)
        HIP_TRY(hipStreamWaitEvent(g.stream, g.reader[r], 0));
        g.reader[r] = nullptr;
    }
    HIP_TRY(hipEventRecord(g.t0[r], g.stream));
    HIP_TRY(hipMemcpyAsync(g.d_rgb[r].p, st, g.depth_off + n * 2, hipMemcpyHostToDevice, g.stream));   // colour + depth: one copy (two cost the copy engine a second set-up: 0.061 -> ~0.05 ms, and the host a call)
    HIP_TRY(hipEventRecord(g.t1[r], g.stream));                   // the batch's front end waits for it (lm_launch_pending)
    d->cur_rgb = g.d_rgb[r].p; d->cur_depth = g.d_depth[r];
    d->have_mask[0] = d->have_mask[1] = false;
    d->last_h2d_ms = 0.f;
    d->frame_valid = true;
    const uint64_t before = d->n_submitted;
    const auto tp2 = std::chrono::steady_clock::now();
    rc = slot_begin(d, threshold, class_ids, num_class_ids, g.d_rgb[r].p, g.d_depth[r], d->have_mask, r);
    if (rc) return rc;
    if (d->n_submitted == before + 1) g.used[r] = true;
    const auto tp3 = std::chrono::steady_clock::now();
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    d->host_prof[0] += 1; d->host_prof[1] += secs(tp0, tp1); d->host_prof[2] += secs(tp1, tp2); d->host_prof[3] += secs(tp2, tp3);
    // A full batch goes out at once; a partial one when the GPU is about to run out of work (partial_batch_due); lm_detector_flush /
    // lm_detector_collect launch what is left.  So the batches are as large as the GPU's backlog allows and no larger.
    {   // how fast the frames arrive (moving average of the gap between submits; a pause counts as 10 ms)
        const double t = host_seconds(tp3);
        if (d->last_submit_at > 0.0) {
            float gap = (float)std::min(10.0, (t - d->last_submit_at) * 1e3);
            // one long gap is a pause, not a change of pace: a tight loop that stops to synchronise (the fence between a warm-up and a timed
            // region, a caller that drains the pipeline now and then) must not look like a camera for its next few frames — they would go
            // out one frame per launch, 0.3 ms of GPU time each.  A stream that has really slowed down is told apart within four frames (the average grows by a quarter per frame).
            if (d->submit_gap_ms > 0.f) gap = std::min(gap, 2.f * d->submit_gap_ms);
            d->submit_gap_ms = d->submit_gap_ms > 0.f ? 0.75f * d->submit_gap_ms + 0.25f * gap : gap;
        }
        d->last_submit_at = t;
    }
    if (d->pend_n >= std::max(1, std::min(d->batch_max, kMaxBatch)) || partial_batch_due(d, host_seconds(tp3))) {
        rc = lm_launch_pending(d);
        const double cost = secs(tp3, std::chrono::steady_clock::now());
        d->host_prof[4] += cost;
        d->launch_cost_ms = 0.75f * d->launch_cost_ms + 0.25f * (float)std::min(1.0, cost * 1e3);
        return rc;
    }
    return LM_OK;
}

// 1 when the refinement of the current bank and frame geometry runs on bit planes (k_local_bits), 0 when on the byte strip planes
// (k_local: single-level pyramids have no refinement; LM_BITPLANES=0; lm_detector_set_paths).  Valid after a match.
extern "C" int lm_detector_refines_on_bit_planes(const lm_detector* d) {
    return d && !d->bank_dirty && bits_active(d, 1) ? 1 : 0;
}

extern "C" int lm_detector_set_paths(lm_detector* d, int refine, int coarse) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    if (refine < 0 || refine > 2 || coarse < 0 || coarse > 1) return lm_set_error(LM_ERR_INVALID, "refine must be 0 (bit planes), 1 (tiles) or 2 (per candidate), coarse 0 (bit planes) or 1 (bytes)");
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "frames in flight: collect them first");
    int rc = lm_launch_pending(d);
    if (rc) return rc;
    d->refine_mode = refine; d->coarse_mode = coarse;
    return LM_OK;
}

extern "C" int lm_detector_set_direct_bits(lm_detector* d, int on) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "frames in flight: collect them first");
    int rc = lm_launch_pending(d);
    if (rc) return rc;
    d->fe_direct = on != 0;
    d->fe_keep_top = (on & 2) != 0;  // tests: the pair stream stays readable after the match (lm_detector_read_stage kind 5) and is cleared before the next frame instead
    d->fe_top_mode = (on & 4) ? 1 : ((on & 8) ? 2 : 0);   // tests: 4 = the OR-ing writer of the pair stream also where whole bytes / dwords could be stored, 8 = no pixel tiles (the whole-dword writer where the geometry allows it)
    return LM_OK;
}

extern "C" int lm_detector_get_paths(const lm_detector* d, int* refine, int* coarse) {
    if (!d || !refine || !coarse) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (d->bank_dirty) return lm_set_error(LM_ERR_INVALID, "no match yet: the paths follow from the bank and the frame geometry");
    const bool bits = bits_active(d, 1);
    *refine = bits ? 0 : (d->geom.levels >= 2 && tiles_wanted(d) && tile_plan_possible(d->geom) ? 1 : 2);
    *coarse = cbits_active(d, 1) ? 0 : 1;
    return LM_OK;
}

extern "C" int lm_detector_flush(lm_detector* d) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    return lm_launch_pending(d);
}

extern "C" int lm_detector_set_batch(lm_detector* d, int frames) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    if (frames < 1 || frames > kMaxBatch) return lm_set_error(LM_ERR_INVALID, "frames per launch must be in [1, %d]", kMaxBatch);
    int rc = lm_launch_pending(d);
    if (rc) return rc;
    d->batch_max = frames;
    return LM_OK;
}

extern "C" int lm_detector_get_batch(const lm_detector* d) { return d ? d->batch_max : 0; }

extern "C" int lm_detector_host_profile(lm_detector* d, double* out8, int reset) {
    if (!d || !out8) return lm_set_error(LM_ERR_INVALID, "null argument");
    for (int i = 0; i < 8; ++i) { out8[i] = d->host_prof[i]; if (reset) d->host_prof[i] = 0; }
    return LM_OK;
}

extern "C" int lm_detector_set_async_collect(lm_detector* d, int on) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    d->async_collect = on != 0;       // frames already launched keep what they were launched with
    return LM_OK;
}

extern "C" int lm_detector_set_batch_queue(lm_detector* d, int batches) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null detector");
    if (batches < 0 || batches > lm_detector::kSlots) return lm_set_error(LM_ERR_INVALID, "batches queued on the GPU must be in [0, %d]", lm_detector::kSlots);
    d->keep_queued = batches;
    return LM_OK;
}

extern "C" int lm_detector_collect(lm_detector* d, int sort_unique, lm_match** out, size_t* n_out) {
    if (!d || !out || !n_out) return lm_set_error(LM_ERR_INVALID, "null argument");
    *out = nullptr; *n_out = 0;
    int rc = lm_collect_frame(d, sort_unique, out, n_out);
    if (rc == 1)
        return lm_set_error(LM_ERR_OVERFLOW, "candidate buffer overflow: capacity raised to %u, submit the frame again "
                            "(lm_detector_match_resident does this by itself)", d->cand_cap);
    return rc;
}

extern "C" int lm_detector_match_resident(lm_detector* d, float threshold, const char* const* class_ids, int num_class_ids,
                                          int sort_unique, lm_match** out, size_t* n_out) {
    if (!d || !out || !n_out) return lm_set_error(LM_ERR_INVALID, "null argument");
    *out = nullptr; *n_out = 0;
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "frames in flight: collect them first");
    for (;;) {   // one pass normally; grow-and-rerun when a buffer overflowed
        int rc = lm_submit_frame(d, threshold, class_ids, num_class_ids);
        if (rc) return rc;
        rc = lm_collect_frame(d, sort_unique, out, n_out);
        if (rc != 1) return rc;
    }
}

extern "C" int lm_detector_match(lm_detector* d, const uint8_t* rgb, const uint16_t* depth, int width, int height, float threshold,
                                 const char* const* class_ids, int num_class_ids, const uint8_t* const* masks, lm_match** out,
                                 size_t* n) {
    int rc = lm_detector_set_frame(d, rgb, depth, width, height, masks);
    if (rc) return rc;
    return lm_detector_match_resident(d, threshold, class_ids, num_class_ids, 1, out, n);
}

extern "C" int lm_detector_last_timings(const lm_detector* d, lm_timings* t) {
    if (!d || !t) return lm_set_error(LM_ERR_INVALID, "null argument");
    *t = d->timings;
    return LM_OK;
}

extern "C" int64_t lm_detector_read_stage(lm_detector* d, int level, int kind, uint8_t* dst, int64_t capacity) {
    if (!d || level < 0 || level >= d->pyramid_levels || kind < 0 || kind > 5) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (d->fW <= 0) return lm_set_error(LM_ERR_INVALID, "no frame processed yet");
    const LevelBufs& b = d->lvl[level];
    const LevelGeom& lv = d->geom.lv[level];
    const uint8_t* src = nullptr;
    int64_t size = 0;
    switch (kind) {
        case 0: src = b.ang.p; size = (int64_t)b.W * b.H; break;
        case 1: src = b.nrm.p; size = (int64_t)b.W * b.H; break;
        case 2: src = d->lm_arena[d->last_arena].p + lv.lm_off[0]; size = (int64_t)8 * lv.T * lv.T * lv.Wd * lv.Hd; break;
        case 3: src = d->lm_arena[d->last_arena].p + lv.lm_off[1]; size = (int64_t)8 * lv.T * lv.T * lv.Wd * lv.Hd; break;
        case 4:       // strip records of a level below the top, colour block then normal block (what k_local_bits reads)
            if (level == d->pyramid_levels - 1) return lm_set_error(LM_ERR_INVALID, "the top level has no strip records");
            src = d->bits_arena[d->last_arena].p + (lv.sm_off[0] >> 1); size = (int64_t)2 * 8 * lv.T * lv.T * lv.NS * lv.Hd * 8; break;
        default:      // pair stream of the top level (what k_coarse_bits reads)
            if (level != d->pyramid_levels - 1) return lm_set_error(LM_ERR_INVALID, "only the top level has a pair stream");
            src = d->cbits_arena[d->last_arena].p; size = (int64_t)d->cbits_npairs * 8; break;
    }
    if (dst && capacity > 0) {
        if (hipSetDevice(d->device) != hipSuccess) return lm_set_error(LM_ERR_HIP, "hipSetDevice failed");
        (void)hipStreamSynchronize(d->stream);
        (void)hipStreamSynchronize(d->mstream);
        if ((kind == 2 || kind == 3) && (level == d->pyramid_levels - 1 ? !d->fe_bytes_top : !d->fe_bytes_low)) {
            // the last front end wrote this level's bit planes only: build its byte planes now, from the quantised maps it left
            if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "frames in flight: collect them first");
            const LevelBufs& q = d->level_bufs(0, level);
            const lm_detector::Slot& sl = d->slot[d->last_arena];
            const bool strips = level < d->pyramid_levels - 1;
            const uint8_t* quant[2] = {q.ang.p, q.nrm.p};
            const uint8_t* mask[2] = {sl.have_mask[0] ? b.mask[0].p : nullptr, sl.have_mask[1] ? b.mask[1].p : nullptr};
            uint8_t* lmp[2] = {d->lm_arena[d->last_arena].p + lv.lm_off[0], d->lm_arena[d->last_arena].p + lv.lm_off[1]};
            uint8_t* smp[2] = {strips ? d->sm_arena[d->last_arena].p + lv.sm_off[0] : nullptr, strips ? d->sm_arena[d->last_arena].p + lv.sm_off[1] : nullptr};
            launch_build_lm(quant, mask, lmp, smp, q.W, q.H, lv.T, d->stream);
            (void)hipStreamSynchronize(d->stream);
        }
        hipError_t e = hipMemcpy(dst, src, (size_t)std::min(size, capacity), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return lm_set_error(LM_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    }
    return size;
}
