// Private view of the detector shared by detector.cpp and pipeline.cpp (not part of the C ABI).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/amd_linemod.h"
#include "host_templates.h"
#include "knobs.h"
#include "lm_kernels.h"

int lm_set_error(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return lm_set_error(LM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                                  __FILE__, __LINE__);                                        \
    } while (0)


namespace lm {}
using namespace lm;

// ---- device buffer helper ---------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return LM_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        HIP_TRY(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        return LM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct LevelBufs {
    int W = 0, H = 0;
    DevBuf<uint8_t> rgb;      // level>0 only (level 0 aliases the frame)
    DevBuf<float> mag;
    DevBuf<uint8_t> ang;      // one-hot quantised orientation (unmasked)
    DevBuf<uint8_t> nrm;      // one-hot quantised normal (unmasked)
    DevBuf<uint8_t> mask[2];  // per modality, optional
};

struct lm_detector {
    // parameters (LL.cpp:1663-1692 + modality defaults :645-650, :968-974)
    int num_features = 63;
    std::vector<int> T_at_level{5, 8};
    int pyramid_levels = 2;
    float weak_threshold = 10.0f, strong_threshold = 55.0f;
    int distance_threshold = 2000, difference_threshold = 50, extract_threshold = 2;
    TemplatesMap class_templates;

    // device
    int device = 0;
    hipStream_t stream = nullptr;       // frame upload / select + front end (and addTemplate)
    hipStream_t mstream = nullptr;      // refinement (a lone frame: all three matching kernels; + the pipeline's NMS / ICP): frame k, while `stream` prepares frame k+1
    hipEvent_t ev[8] = {};
    int shard_rank = 0, shard_world = 1;

    // frame
    int fW = 0, fH = 0;
    bool frame_valid = false, have_mask[2] = {false, false};
    DevBuf<uint8_t> frame_rgb;
    DevBuf<uint16_t> frame_depth;
    const uint8_t* cur_rgb = nullptr;           // what the front end reads: frame_rgb / frame_depth, or — for a frame that came in through
    const uint16_t* cur_depth = nullptr;        // lm_detector_submit_frame — the ingest ring's device buffers (no device-to-device copy)
    DevBuf<uint8_t> nrm_raw;                    // normals before the median (level 0)
    static constexpr int kSlots = 16;           // frames in flight (lm_detector_submit / collect; lm_detector_max_in_flight): every one owns
                                                // its arenas, candidate / record buffers and pinned result memory (~35 MB at VGA)
    DevBuf<uint8_t> lm_arena[kSlots], sm_arena[kSlots];   // linear memories per result slot: the front end of frame k+1 writes one set
                                                // while the matching kernels of frame k read the other
    int last_arena = 0;                         // set written by the most recent front end (lm_detector_read_stage)

    LevelBufs lvl[kMaxLevels];                  // front-end intermediates of frame 0 of a batch (and of addTemplate, read_stage)
    LevelBufs lvl_x[kMaxBatch - 1][kMaxLevels]; // ... of frames 1.. of a batch (allocated on first use)
    DevBuf<uint8_t> nrm_raw_x[kMaxBatch - 1];
    LevelBufs& level_bufs(int b, int l) { return b == 0 ? lvl[l] : lvl_x[b - 1][l]; }
    FrameGeom geom{};
    size_t lm_block_bytes[kMaxLevels] = {};
    std::vector<DevBuf<uint8_t>> slot_rgb;      // frames parked in HBM (lm_detector_store_frame)
    std::vector<DevBuf<uint16_t>> slot_depth;
    std::vector<int> slot_w, slot_h;
    void* pinned = nullptr;          // staging for H2D frame and D2H results
    size_t pinned_bytes = 0;
    float last_h2d_ms = 0.f;

    // bank on device
    bool bank_dirty = true;
    int bank_geom_W = -1, bank_geom_H = -1;
    std::vector<std::string> bank_classes;          // sorted
    std::vector<int> bank_class_base;               // first flat pyramid index per class
    std::vector<int> bank_class_count;
    std::vector<TemplEntry> h_entries;
    DevBuf<TemplEntry> d_entries;
    DevBuf<int32_t> d_feat_off;
    DevBuf<uint32_t> d_feat_xy;
    DevBuf<uint32_t> d_feat_word, d_run_mask;       // refinement levels: feature words + class-run masks (lm_kernels.h)
    // work list
    std::vector<int32_t> work_pyr;
    std::shared_ptr<std::vector<int32_t>> work_cls, work_tid;   // shared with in-flight result slots
    DevBuf<int32_t> d_work;
    std::vector<std::string> work_key;              // class_ids the cached work list was built for
    int work_key_rank = -1, work_key_world = -1;
    bool work_valid = false;
    int64_t work_coarse_bytes = 0;
    DevBuf<Candidate> d_cands;
    DevBuf<Candidate> d_matches_dev;                // HBM copy of the refined records (on-device NMS / top-K, duplicate removal)
    DevBuf<unsigned long long> d_hash;              // open-addressing table of k_dedupe
    DevBuf<TileRec> d_tiles;                        // per result slot: the tiles k_coarse planned (cand_cap / 2 records each)
    DevBuf<uint8_t> d_todo;                         // per result slot: 1 = candidate that no tile serves (refined on its own)
    bool use_tiles = true;                          // LM_TILES=0: every candidate on its own (the round-1 refinement)
    bool reference_order = false;                   // lm_detector_set_reference_order / LM_REFERENCE_ORDER=1: match() returns the reference's own permutation (sort_unique 3)
    DevBuf<ulonglong2> d_distinct_keys;             // the distinct records as 128-bit sort keys, per result slot (multi-GPU exchange)
    DevBuf<int32_t> d_work_cls, d_work_tid;         // class position / template id per work item
    DevBuf<unsigned long long> d_counters;          // kCounterWords working counters per result slot: zero between frames (k_dedupe's last block resets them)
    DevBuf<unsigned long long> d_final;             // per result slot: [0] candidates, [1] distinct, [2] alive, [3] key overflow of the finished frame
    uint32_t cand_cap = 1u << 18;                   // candidate capacity wanted (raised when a frame overflowed)
    uint32_t buf_cand_cap = 0;                      // ... the per-slot device buffers are laid out for (frames in flight use this one)
    // Result slots: the refinement kernel of a later frame writes into one pinned buffer while the host
    // collects an earlier frame from another (lm_detector_submit / lm_detector_collect).
    struct Slot {
        Candidate* h_matches = nullptr;             // pinned: the raw records, copied from matches_dev when a collect asks for them
        Candidate* h_distinct = nullptr;            // pinned: the same without exact duplicates (k_dedupe), unordered
        uint32_t match_cap = 0;
        unsigned long long* h_counters = nullptr;   // pinned: [0] candidate count, [1] distinct records, [2] records alive, [8..] 2 words of statistics per refinement block
        // A BATCH of consecutive slots is launched together (one front end / coarse / refinement / duplicate-removal launch for all
        // its frames); the events below are those of the batch's first slot (`leader`).
        int leader = -1, batch_n = 1;               // first slot of the batch this frame was launched with, frames in that batch
        bool launched = false;                      // false: submitted but still waiting for its batch to fill (lm_detector_flush / collect launch it)
        const uint8_t* in_rgb = nullptr;            // the frame the front end reads: the detector's resident frame or the slot's ingest ring entry
        const uint16_t* in_depth = nullptr;
        bool have_mask[2] = {false, false};         // lone frames only (Detector.match with masks)
        int ring = -1;                              // ingest ring entry holding the frame (-1: resident frame)
        // prepared by the collector thread (streamed frames): the frame's Detector::match list, ready when `ready` != 0
        bool prep_queued = false;                   // a collector job covers this frame
        std::atomic<int> ready{0};                  // 0 not yet, 1 list prepared, 2 nothing prepared (overflow: the caller's path decides)
        lm_match* prep = nullptr;
        size_t prep_n = 0;
        float prep_collect_ms = 0.f, prep_merge_ms = 0.f;
        hipEvent_t ev[6] = {};                      // stage timing: front end 0 -> 1, coarse 1 -> 3, refinement 3 -> 4 (2 and 5: not recorded since round 5)
        hipEvent_t done = nullptr;                  // recorded after the batch's last kernel: the only event the host waits on
        hipEvent_t fe_done = nullptr;               // front end of this slot finished (eager, on `stream`): `mstream` waits for it
        bool pending = false;
        float threshold = 0.f, h2d_ms = 0.f;
        int num_work = 0;
        uint32_t cand_cap = 0;                      // the candidate capacity / buffer this frame was submitted with (d->cand_cap may grow before it is collected)
        const Candidate* cands = nullptr;
        const Candidate* matches_dev = nullptr;     // the frame's refined records in HBM (one per candidate, work = -1: dropped)
        int64_t coarse_bytes = 0;
        std::shared_ptr<std::vector<int32_t>> work_cls, work_tid;
        std::chrono::steady_clock::time_point t0, t1;
    } slot[kSlots];
    // Bit planes (match.hip): both matching kernels read 1-bit response planes by default.
    DevBuf<uint8_t> cbits_arena[kSlots];            // pair stream of the top level's flat memories (k_coarse_bits)
    uint32_t cbits_byte0 = 0, cbits_npairs = 0;     // ... = arena bytes [byte0, byte0 + 32 npairs): the top level's two blocks with their zero tails
    DevBuf<uint8_t> bits_arena[kSlots];             // strip records of the levels below the top (the strip arena's layout at half the offsets; k_local_bits)
    int bits_max_nf = 0, cbits_max_nf = 0;          // largest feature count of a template entry below the top level / at the top level (which counter width the kernels need)
    bool bits_all_in = false;                       // every candidate's windows lie inside their planes at every level: k_local_bits leaves nothing to k_local
    // Which kernels match() uses (lm_detector_set_paths; tests and measurements of the byte paths).  refine: 0 = bit planes when the pyramid has a
    // level below the top (default), 1 = byte strip planes with tiles (round 2-3's k_local), 2 = byte planes, every candidate on its own
    // (round 1).  coarse: 0 = pair stream when the refinement runs on bit planes (default), 1 = byte linear memories (k_coarse).
    int refine_mode = 0, coarse_mode = 0;
    bool fe_direct = true;                          // lm_detector_set_direct_bits: the front end writes bit planes directly where nothing reads the bytes (0: bytes + k_pack_bits / k_pack_top)
    bool fe_keep_top = false;                       // lm_detector_set_direct_bits(d, 2)
    int fe_top_mode = 0;                            // lm_detector_set_direct_bits(d, 4 / 8): which writer of the pair stream (fe_job_top_bits)
    bool fe_bytes_low = true, fe_bytes_top = true;  // did the last front end write the byte planes of the levels below the top / of the top level (read_stage builds them on demand otherwise)
    hipEvent_t resident_reader = nullptr;           // front end (event of its batch) that reads the resident frame buffers: lm_detector_select_frame's copy waits for it
    bool cbits_clean[kSlots] = {};                  // the slot's pair stream is all zero (what the front end's OR-ing writer needs)
    uint64_t n_submitted = 0, n_collected = 0, n_launched = 0;
    // frames submitted but not launched yet: slots pend_first .. pend_first + pend_n - 1 (modulo kSlots), same threshold and work list
    int batch_max = 8;                              // frames per launch in stream mode (lm_detector_set_batch, LM_FRAME_BATCH; <= kMaxBatch).  8 since round 4: 0.074 against 0.089 ms per frame over 200 steps, the same at 20 (profiles/r04_stream_ab.txt)
    int pend_first = 0, pend_n = 0;
    float pend_threshold = 0.f;
    // Launched batches the GPU may still be working on, oldest first.  When does a streamed frame that does not fill its batch go
    // out?  The GPU must never wait for a batch to fill, and a batch launched early is a small one (a lone frame costs ~2.5x its
    // share of a batch of four).  The detector therefore keeps an estimate of when the GPU will have finished what it was given
    // (`gpu_free_at`: launch times + the measured duration of batches of each size, corrected whenever a collect sees a batch
    // finish) and launches a partial batch when that moment is closer than `launch_slack_ms` — the enqueue latency of a batch —
    // either at a submit or just before a collect blocks.  Without a measured duration yet the rule is the backlog in batches:
    // launch while fewer than `keep_queued` are unfinished.  keep_queued = 0 switches both off (batches of exactly batch_max frames;
    // flush / collect launch what is left).
    struct QueuedBatch {
        uint64_t first_frame;                       // index of the batch's first frame
        int slot, frames;                           // its leader slot, its size
        double launched_at;                         // host clock (seconds, steady_clock)
        bool gpu_idle_at_launch;                    // nothing unfinished before it: it started when it was launched
    };
    std::vector<QueuedBatch> queued;
    int keep_queued = 2;                            // LM_BATCH_QUEUE
    float batch_ms[kMaxBatch + 1] = {};             // measured GPU time of a batch of n frames (moving average); 0 = not seen yet
    double gpu_free_at = 0.0;                       // estimate, host clock
    double last_done_at = -1.0;                     // when the most recently collected batch finished, if a collect saw it happen (-1: unknown)
    uint64_t last_done_end = 0;                     // index of the frame after that batch
    float launch_slack_ms = 0.15f;                  // LM_LAUNCH_SLACK_US
    double last_submit_at = 0.0;                    // lm_detector_submit_frame: host clock of the previous call,
    float submit_gap_ms = 0.f;                      // moving average of the gap between calls (0 = no second call yet: treated as sparse),
    float launch_cost_ms = 0.1f;                    // and of the host time one lm_launch_pending takes
    // Helper threads of the streamed path (one pool per detector, started with the first job; LM_HOST_THREADS, default 3, 0 = none).  They
    // never call into the HIP runtime — a second thread inside the runtime made every HIP call of the calling thread take ~0.1 ms (round 3:
    // the collector thread that waited on events) — and do two things:
    //   * the staging copy of a streamed frame (1.5 MB at VGA: 40 us on one core) in slices, beside the caller's own slice;
    //   * the canonical result lists: frames are launched in batches, so when the collect of a batch's first frame has seen the batch's
    //     event, the records of ALL its frames are in pinned memory — the helpers convert, sort and unique the lists of the later frames
    //     (30-35 us each at 2k templates) while the caller's thread does the first one; the later collects only hand their lists over.
    // A helper spins for ~0.3 ms after a job (a stream hands them one every few tens of us) and sleeps on the condition variable otherwise.
    struct HostPool {
        std::vector<std::thread> th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::function<void()>> jobs;
        std::atomic<int> posted{0};                 // jobs in the queue
        std::atomic<int> asleep{0};                 // helpers waiting on cv
        bool stop = false, started = false;
        int threads = 3;
    } pool;
    // host-side wall time of the streamed path, accumulated (lm_detector_host_profile): [0] frames, [1] staging copy, [2] H2D enqueue,
    // [3] slot bookkeeping, [4] batch launches, [5] collect: waiting for the GPU, [6] record conversion, [7] canonical sort + unique
    double host_prof[8] = {};
    uint64_t ncand_hint = 0;                        // coarse candidates of the last collected frame: sizes k_dedupe's grid
    bool early_batch_used = false;                  // the burst's one early (partial) batch of a tight stream has gone out (partial_batch_due)
    bool async_collect = true;                      // lm_detector_set_async_collect / LM_ASYNC_COLLECT=0: the lists of a batch's later frames are prepared by the helper threads

    // Live-stream ingest (lm_detector_submit_frame): one ring entry per result slot.  The host frame is staged in the entry's
    // pinned buffer (or written there by the caller: lm_detector_ingest_buffer), copied to the entry's device buffers on a
    // dedicated copy stream — the H2D of frame k+1 runs beside the front end of frame k and the matching kernels of k-1 —
    // and the front end reads it there.  An entry is free again when its frame was collected.
    struct Ingest {
        hipStream_t stream = nullptr;
        void* pinned[kSlots] = {};
        size_t pinned_bytes[kSlots] = {};
        DevBuf<uint8_t> d_rgb[kSlots];                 // colour image, then (16-byte aligned) the depth image: ONE upload per frame, the pinned entry has the same layout
        uint16_t* d_depth[kSlots] = {};                // = d_rgb + depth_off
        size_t depth_off = 0;
        hipEvent_t t0[kSlots] = {}, t1[kSlots] = {};   // timing of the H2D (copy stream)
        bool used[kSlots] = {};                        // the slot's frame came in through the ring (lm_timings.h2d_ms from t0/t1)
        hipEvent_t reader[kSlots] = {};                // front end (of another slot: a resident re-match of the streamed frame) that still reads the entry
    } ingest;

    // multi-GPU exchange on the device (exchange.cpp): its own stream, so that the sort of frame k's records, the caller's
    // RCCL all-gather and the merge run beside the matching kernels of frame k+1
    struct Exchange {
        hipStream_t stream = nullptr;               // created with the detector; also runs the duplicate removal of pipelined frames
        DevBuf<int32_t> d_merged[kSlots];
        DevBuf<ulonglong2> d_runs;                  // scratch of the per-rank sort
        int32_t* h_merged[kSlots] = {};             // pinned
        size_t h_words[kSlots] = {};
        hipEvent_t done[kSlots] = {};
        int state[kSlots] = {};                     // 0 idle, 1 packed, 2 merged (result on its way to h_merged)
        int cap[kSlots] = {}, world[kSlots] = {};
        int done_slot[kSlots] = {};                 // the slot whose `done` event covers this frame's group
    } xchg;
    int local_blocks = 0;
    int num_cus = 256;

    // template extraction on the device (train.hip): scratch of the view being prepared + per-view candidate lists of a chunk
    struct Train {
        DevBuf<uint8_t> mask[kMaxLevels], lab[kMaxLevels], user_mask;
        DevBuf<int32_t> hrun[kMaxLevels];
        DevBuf<unsigned long long> keys;
        DevBuf<uint32_t> counts;
        DevBuf<int32_t> bbox, out;
    } train;

    bool fe_fused = true;            // addTemplate's front end: independent jobs share a launch (k_fe_stage); LM_FE_FUSED=0: one launch per job

    lm_timings timings{};
};


// detector.cpp
int lm_submit_frame(lm_detector* d, float threshold, const char* const* class_ids, int num_class_ids);
int lm_collect_frame(lm_detector* d, int sort_unique, lm_match** out, size_t* n_out);   // sort_unique < 0: discard the records
int lm_launch_pending(lm_detector* d);                                                  // launches the frames waiting for their batch to fill

// A writer on the frame stream (`stream`) — upload, frame selection with a geometry change, training — touches level buffers, arenas and the
// resident frame that a batch in flight on the matching stream reads.  Stream order only runs the other way (order_after_default_stream: batch
// after `stream`), so these entry points must find nothing launched and uncollected; they check it at run time and refuse.  A DIAG build
// (make DIAG=1) also aborts loudly if a future writer forgets the check.
#ifdef LM_DIAG
#include <stdio.h>
#include <stdlib.h>
#define LM_DIAG_IDLE(d, what)                                                                                                    \
    do {                                                                                                                         \
        if ((d)->n_launched != (d)->n_collected) {                                                                               \
            fprintf(stderr, "LM_DIAG: %s writes on the frame stream with %llu batches' frames launched and %llu collected\n", (what), \
                    (unsigned long long)(d)->n_launched, (unsigned long long)(d)->n_collected);                                 \
            abort();                                                                                                             \
        }                                                                                                                        \
    } while (0)
#else
#define LM_DIAG_IDLE(d, what) do { } while (0)
#endif
