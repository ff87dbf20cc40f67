/*
 * amd_linemod.h — C ABI of libamdlinemod.so, the MI355X (gfx950) implementation of the
 * linemodLevelup hot path (LINE-MOD template matching + point-to-plane ICP pose refinement).
 *
 * This is the drop-in boundary for the reference's pybind11 module `linemodLevelup_pybind`
 * (/root/reference/linemodLevelup/pybind11.cpp:7-35).  Every entry point below names the reference
 * interface it replaces ("pybind11.cpp:N" = binding, "LL.cpp:N"/"LL.h:N" = linemodLevelup.{cpp,h}).
 * Plain pointers and sizes only; no torch / OpenCV / STL types.  The Python wrapper
 * 6dpose_amd/linemodLevelup_pybind.py and bench.py are the only intended callers; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every `int` function returns LM_OK (0) or a negative LM_ERR_* code unless stated otherwise;
 *     lm_last_error() returns a thread-local message (the reference raises cv::Exception ->
 *     Python RuntimeError for the same preconditions, e.g. LL.cpp:1136,1217-1218,1291,1707).
 *   - images are C-contiguous host buffers borrowed for the duration of the call:
 *     rgb  uint8  [height][width][3]  (channel order as given, LL.cpp:391-412 is order-sensitive)
 *     depth uint16 [height][width]    (millimetres)
 *     mask uint8  [height][width]     (non-zero = valid), may be NULL
 *   - a detector owns one HIP stream on one device; handles are not re-entrant.
 *   - there is NO CPU fallback: creation fails (LM_ERR_NO_DEVICE) when no gfx950 device is visible.
 */
#ifndef AMD_LINEMOD_H
#define AMD_LINEMOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LM_OK 0
#define LM_ERR_INVALID (-2)     /* bad argument / violated reference precondition (CV_Assert) */
#define LM_ERR_NO_DEVICE (-3)   /* no HIP device / HIP runtime failure at creation */
#define LM_ERR_HIP (-4)         /* HIP runtime error during a call */
#define LM_ERR_IO (-5)          /* file could not be read / written / parsed */
#define LM_ERR_NOT_FOUND (-6)   /* unknown class id */
#define LM_ERR_OVERFLOW (-7)    /* pipelined mode only: candidate buffer overflow, capacity raised, resubmit the frame */

typedef struct lm_detector lm_detector;

/* linemodLevelup::Match (LL.h:225-258; bound at pybind11.cpp:16-22).  class_index indexes the
 * class list the match call used (caller's class_ids order, or sorted class order when empty). */
typedef struct lm_match {
    int32_t x;
    int32_t y;
    float similarity;
    int32_t class_index;
    int32_t template_id;
} lm_match;

/* Per-stage device time of the last match call, HIP events on the detector's stream (ms). */
typedef struct lm_timings {
    float h2d_ms;        /* frame upload (0 when the frame was already resident)            */
    float frontend_ms;   /* quantise + spread + response + linearise, all levels (A1-A7)     */
    float coarse_ms;     /* similarity over all positions at the top level + threshold (A8-A10) */
    float local_ms;      /* 16x16 refinement down the pyramid (A11)                          */
    float d2h_ms;        /* match download                                                   */
    float total_ms;      /* first event to last event                                        */
    int64_t coarse_candidates;
    int64_t local_evals;
    int64_t matches_pre_unique;
    int64_t templates;   /* template pyramids searched by this rank                          */
    int64_t coarse_bytes;/* algorithmic response bytes read by the coarse pass  */
    int64_t local_bytes; /* algorithmic response bytes read by the local pass                */
    float host_submit_ms;  /* host wall time: call entry -> everything enqueued               */
    float host_wait_ms;    /* host wall time blocked in the stream synchronisation            */
    float host_collect_ms; /* host wall time: read-back bookkeeping + record conversion       */
    float host_merge_ms;   /* host wall time: canonical sort + unique (0 when not requested)  */
    int32_t batch_frames;  /* frames served by the launches the stage times above belong to (1 for a lone frame;
                            * lm_detector_submit_frame batches): per-frame device time = stage time / batch_frames */
} lm_timings;

const char *lm_last_error(void);
const char *lm_version(void);
/* Number of visible HIP devices (0 when none / runtime unavailable). */
int lm_device_count(void);
/* Binds the CALLING THREAD (and the threads it starts afterwards) to the CPUs local to `device` (sysfs local_cpulist of its PCI
 * function), within what the thread was allowed before; cpulist_out (optional) receives the list.  Call it before creating the
 * detector: the pinned staging buffers of lm_detector_submit_frame then live on the GPU's NUMA node.  A deployment that already
 * runs under numactl / a cpuset needs nothing.  No reference counterpart (the reference has no device). */
int lm_bind_thread_near_device(int device, char *cpulist_out, size_t cap);

/* ---- Detector -------------------------------------------------------------------------------
 * Detector(), Detector(T), Detector(num_features, T): pybind11.cpp:26-28, LL.cpp:1663-1692.
 * num_features <= 0 selects the default 63; T==NULL selects {5,8}.  `device` is the HIP ordinal. */
int lm_detector_create(int num_features, const int *T, int num_levels, int device, lm_detector **out);
void lm_detector_destroy(lm_detector *d);

/* Detector::addTemplate (pybind11.cpp:29, LL.cpp:1943-1975).  Returns the new template_id (>=0),
 * -1 when no valid template could be extracted (the reference's own convention, LL.cpp:1964-1966),
 * or an LM_ERR_* code (< -1).  Quantisation runs on the GPU; so does the greedy feature selection when a mask is given
 * (train.hip, checked against the reference's golden YAML), on the host for the maskless call. */
int lm_detector_add_template(lm_detector *d, const uint8_t *rgb, const uint16_t *depth, const uint8_t *mask,
                             int width, int height, const char *class_id);

/* Detector::readClasses / writeClasses for ONE class (pybind11.cpp:30-31; LL.cpp:2043-2146):
 * OpenCV FileStorage YAML 1.0 schema; `.gz` paths are NOT supported by this library (the Python
 * wrapper decompresses).  read returns LM_ERR_INVALID on the reference's CV_Asserts (modalities,
 * pyramid_levels, consecutive template ids, class already present). */
int lm_detector_read_class(lm_detector *d, const char *path, const char *class_id_override);
int lm_detector_write_class(lm_detector *d, const char *class_id, const char *path);

/* Detector::write / Detector::read (LL.cpp:2013-2041; C++-only in the reference): pyramid_levels, T and the parameters of
 * the two modalities (ColorGradient::write :686-692, DepthNormal::write :1012-1020) as OpenCV FileStorage YAML.  read clears
 * the classes (LL.cpp:2015). */
int lm_detector_write_params(const lm_detector *d, const char *path);
int lm_detector_read_params(lm_detector *d, const char *path);

/* Bulk import of a packed class (binary bank for >=16k templates).
 * features: [total][3] int32 (x,y,label); tmpl_offsets: [num_pyramids*levels*2+1] prefix offsets in
 * reference TemplatePyramid order (level-major, modality-minor; LL.h:336-337); tmpl_wh:
 * [num_pyramids*levels*2][2] width,height. */
int lm_detector_add_class_packed(lm_detector *d, const char *class_id, int num_pyramids,
                                 const int32_t *features, const int32_t *tmpl_offsets, const int32_t *tmpl_wh);

/* Packed binary bank file next to the reference's one-YAML-per-class files (LL.cpp:2124-2146): the same
 * information — class id, per template width/height/pyramid level, per feature x, y, label (LL.cpp:2072-2122) — as flat
 * arrays, 4 bytes per feature (x int16, y int13, label 3 bits), read through mmap.  write: the named classes, or all of
 * them when num_class_ids == 0; LM_ERR_INVALID when a feature does not fit the packing (the YAML path has no such
 * limit).  read: adds the named classes of the file (all when num_class_ids == 0), so a rank that serves some of the
 * objects touches only their part of the file; pyramid_levels must match (LL.cpp:2052), a class already present is an
 * error (LL.cpp:2059) and then nothing is added.  lm_bank_file_info / _class_id inspect a file without a detector. */
int lm_detector_write_bank(const lm_detector *d, const char *path, const char *const *class_ids, int num_class_ids);
int lm_detector_read_bank(lm_detector *d, const char *path, const char *const *class_ids, int num_class_ids);
int lm_bank_file_info(const char *path, int32_t *pyramid_levels, int32_t *num_classes, int64_t *num_pyramids, int64_t *num_features);
int lm_bank_file_class_id(const char *path, int index, char *out, int capacity);

/* Detector::numClasses / classIds / numTemplates (LL.h:343, LL.cpp:1984-2011). */
int lm_detector_num_classes(const lm_detector *d);
const char *lm_detector_class_id(const lm_detector *d, int index);   /* sorted (std::map) order */
int lm_detector_num_templates(const lm_detector *d, const char *class_id); /* NULL = all classes */
int lm_detector_pyramid_levels(const lm_detector *d);
int lm_detector_get_T(const lm_detector *d, int level);

/* Detector::getTemplates (pybind11.cpp:34, LL.cpp:1976-1982): entry `index` (0..levels*2-1) of
 * template pyramid `template_id`.  Writes width/height/pyramid_level/num_features; copies at most
 * `capacity` features (x,y,label triples) into `features` (may be NULL to query the count). */
int lm_detector_get_template(const lm_detector *d, const char *class_id, int template_id, int index,
                             int32_t *width, int32_t *height, int32_t *pyramid_level,
                             int32_t *num_features, int32_t *features, int capacity);

/* Multi-GPU sharding: this rank searches the contiguous slice
 * [N*rank/world, N*(rank+1)/world) of the N template pyramids a match call selects; template ids
 * stay global.  Default (0,1). */
int lm_detector_set_shard(lm_detector *d, int rank, int world);

/* Multi-GPU exchange on the device.  Detector::match sorts and adjacent-uniques the records of ALL templates
 * (LL.cpp:1771-1776); with the bank sharded over W ranks that is a distributed merge sort, done without the host:
 *   lm_detector_submit(...)                         this rank's shard of the frame
 *   lm_detector_exchange_pack(d, send, capacity)    sorts the rank's distinct records into `send` (device memory,
 *                                                   lm_exchange_block_bytes(capacity) bytes: header + a sorted run of keys)
 *   all-gather of the W blocks into `recv`          the caller's collective (RCCL), enqueued on lm_detector_exchange_stream(d)
 *   lm_detector_exchange_merge(d, recv, W, capacity) merges the W runs, marks what std::unique drops, copies to pinned memory
 *   lm_detector_exchange_collect(d, &out, &n, &failed) waits for the oldest frame in flight: the same list on every rank,
 *                                                   identical to an unsharded lm_detector_match (lm_free(out))
 * pack / merge apply to the most recently submitted frame and only enqueue work on the exchange stream (hipStream_t
 * returned as void*); up to lm_detector_max_in_flight() frames can be in flight.  capacity: power of two in [256, lm_exchange_max_capacity()] (65536), the same on every rank.
 * *failed != 0 (on every rank alike, *out == NULL): > 0 some rank had that many distinct records (> capacity), < 0 a
 * candidate buffer overflowed or a field did not fit the key — rerun the frame through lm_detector_match_resident +
 * lm_merge_matches (sharded.py does). */
int lm_exchange_max_capacity(void);
int lm_detector_max_in_flight(void);
void *lm_detector_exchange_stream(lm_detector *d);
size_t lm_exchange_block_bytes(int capacity);
int lm_detector_exchange_pack(lm_detector *d, void *send_block, int capacity);
int lm_detector_exchange_merge(lm_detector *d, const void *recv_blocks, int world, int capacity);
/* The same for a frame named by its number in submission order (0, 1, ...): a streamed frame is launched with its batch, so its
 * exchange is enqueued once lm_detector_frames_launched() has passed it (sharded.DeviceExchange does this for every frame, in
 * order, after each submit and before each collect).  frames_submitted() is the number the next submitted frame gets. */
uint64_t lm_detector_frames_submitted(const lm_detector *d);
uint64_t lm_detector_frames_launched(const lm_detector *d);
uint64_t lm_detector_frames_collected(const lm_detector *d);
int lm_detector_exchange_pack_frame(lm_detector *d, uint64_t frame_no, void *send_block, int capacity);
int lm_detector_exchange_merge_frame(lm_detector *d, uint64_t frame_no, const void *recv_blocks, int world, int capacity);
/* One all-gather for a GROUP of frames: every rank sends [frame][block] and receives [rank][frame][block]; recv_blocks = this frame's
 * block of rank 0, rank j's lies j * rank_stride_bytes further. */
int lm_detector_exchange_merge_frame_strided(lm_detector *d, uint64_t frame_no, const void *recv_blocks, int world, int capacity,
                                             size_t rank_stride_bytes);
/* ... and the n frames of a group in one call: frame first + i uses block i of send_blocks / of every rank's part of recv_blocks. */
int lm_detector_exchange_pack_group(lm_detector *d, uint64_t first, int n, void *send_blocks, int capacity);
int lm_detector_exchange_merge_group(lm_detector *d, uint64_t first, int n, const void *recv_blocks, int world, int capacity);
int lm_detector_exchange_collect(lm_detector *d, lm_match **out, size_t *n, int *failed);
/* the same into caller memory (capacity records; world * capacity always suffices): no allocation, one pass */
int lm_detector_exchange_collect_into(lm_detector *d, lm_match *dst, size_t capacity, size_t *n, int *failed);

/* The collective, issued by the library (BASELINE north_star: "RCCL all-gather over xGMI of per-GPU top-K matches"; the caller shape is the
 * reference's C++ driver, linemodLevelup/test.cpp:111-130, one process per GPU).  librccl.so is loaded at run time (dlopen; LM_RCCL_LIB
 * names another file), so single-GPU use needs no RCCL.  Rank 0 calls lm_comm_unique_id and carries the 128 bytes to the other ranks by
 * whatever it has (MPI_Bcast, a file, a socket, torch.distributed's store): the library has no transport of its own.
 *   lm_comm_available()                                1 if librccl.so could be loaded
 *   lm_comm_create(id, rank, world, device, &comm)     ncclCommInitRank on this rank's device (collective: every rank calls it)
 *   lm_exchange_allgather(d, comm, send, recv, bytes)  ncclAllGather of `bytes` per rank on lm_detector_exchange_stream(d): in stream order
 *                                                      after the pack kernels of the frames it carries, before their merges; nothing blocks the host
 *   lm_detector_exchange_group(d, comm, first, n, send, recv, capacity)
 *                                                      pack_group + all-gather + merge_group of frames first .. first + n - 1 in one call
 *                                                      (send: n blocks, recv: world x n blocks of lm_exchange_block_bytes(capacity), device memory) */
typedef struct lm_comm lm_comm;
int lm_comm_available(void);
int lm_comm_unique_id(void *id128);
int lm_comm_create(const void *id128, int rank, int world, int device, lm_comm **out);
void lm_comm_destroy(lm_comm *c);
int lm_comm_rank(const lm_comm *c);
int lm_comm_world(const lm_comm *c);
int lm_exchange_allgather(lm_detector *d, lm_comm *c, const void *send, void *recv, size_t bytes_per_rank);
int lm_detector_exchange_group(lm_detector *d, lm_comm *c, uint64_t first, int n, void *send_blocks, void *recv_blocks, int capacity);

/* Detector::match (pybind11.cpp:32-33, LL.cpp:1702-1777).  class_ids may be NULL/0 = all classes.
 * masks: NULL, or two pointers (colour, depth modality), each NULL or a [height][width] uint8 mask.
 * On success *out is a malloc'd array of *n matches in the canonical order of SURVEY §8a A12
 * (similarity desc, template_id asc, class position asc, y asc, x asc; then adjacent-unique on
 * (x,y,similarity,class)), to be released with lm_free().  With world>1 the result holds only
 * this rank's matches, still sorted+uniqued; lm_merge_matches() merges gathered lists. */
int lm_detector_match(lm_detector *d, const uint8_t *rgb, const uint16_t *depth, int width, int height,
                      float threshold, const char *const *class_ids, int num_class_ids,
                      const uint8_t *const *masks, lm_match **out, size_t *n);

/* The same in two steps, so that a benchmark can time the device path with the frame already
 * resident in HBM: lm_detector_set_frame uploads (and keeps) the frame; lm_detector_match_resident
 * runs front end + matching on it.  `sort_unique`: 1 = canonical sort + unique (Detector::match); 0 = the raw pre-unique
 * records, unordered; 2 = the records without exact duplicates (same x, y, template: several coarse candidates refined to
 * one position; removed on the device, std::unique drops them in any merge), unordered — what the multi-GPU gather ships. */
int lm_detector_set_frame(lm_detector *d, const uint8_t *rgb, const uint16_t *depth, int width, int height,
                          const uint8_t *const *masks);
/* Stream support: park frames in HBM slots once, then make one current with a
 * device-to-device copy on the detector's stream (no host traffic inside a timed region). */
int lm_detector_store_frame(lm_detector *d, int slot, const uint8_t *rgb, const uint16_t *depth, int width, int height);
int lm_detector_select_frame(lm_detector *d, int slot);
int lm_detector_match_resident(lm_detector *d, float threshold, const char *const *class_ids, int num_class_ids,
                               int sort_unique, lm_match **out, size_t *n);
/* Pipelined stream mode: lm_detector_submit enqueues front end + matching of the current
 * frame and returns; lm_detector_collect waits for the OLDEST submitted frame and returns its matches.
 * Up to lm_detector_max_in_flight() (sixteen) frames may be in flight: the front end of frame k+2 and the matching kernels of frame k+1 run on two
 * streams while the host sorts frame k:
 *   select_frame(k+2); submit(); collect() -> frame k; ...
 * lm_detector_match_resident == submit + collect. */
int lm_detector_submit(lm_detector *d, float threshold, const char *const *class_ids, int num_class_ids);
int lm_detector_collect(lm_detector *d, int sort_unique, lm_match **out, size_t *n);
/* sort_unique = 3 (collect, match_resident), or lm_detector_set_reference_order(d, 1) for every sort_unique = 1 call of this
 * detector (lm_detector_match included; LM_REFERENCE_ORDER=1 sets it at creation): the list exactly as the reference's
 * Detector::match returns it — its std::sort (Match::operator<, which ignores x, y) and std::unique (operator==, which ignores
 * template_id) replayed on the pre-unique records in the order the reference appends them (LL.cpp:1753-1776), with the same
 * libstdc++ introsort.  Unlike the canonical order (the default) this keeps the duplicates the reference keeps: on fixture
 * bank 63 at threshold 55 it returns the reference's 560 entries, the canonical order 366 (the same 358 distinct positions).
 * Host-side cost: one extra device-to-host copy of the candidate buffer and a sort of all pre-unique records; single GPU only
 * (a sharded match merges canonically). */
int lm_detector_set_reference_order(lm_detector *d, int on);
/* Live-stream ingest (the per-frame call of linemod_ros/detect.py:83-138 and of the dataset loop
 * linemod_and_levelup_test.py:314-327, which hand a NEW host frame to every Detector.match): lm_detector_submit_frame =
 * "upload this host frame + lm_detector_submit" without blocking.  The frame is staged in a ring of pinned buffers (one per
 * frame in flight), copied to HBM on a dedicated copy stream — the H2D of frame k+1 overlaps the front end of frame k and
 * the matching kernels of frame k-1 — and the call returns as soon as everything is enqueued; rgb / depth are borrowed
 * only until it returns.  Results come back through lm_detector_collect in submission order:
 *   submit_frame(f0); submit_frame(f1); submit_frame(f2); collect() -> f0; submit_frame(f3); collect() -> f1; ...
 * Zero-copy variant: lm_detector_ingest_buffer returns the pinned staging pointers the NEXT lm_detector_submit_frame will
 * use (a camera driver / decoder writes the frame there); passing exactly these pointers skips the staging copy.
 * No masks (the reference's callers pass masks=[]).  The frame size may change only with no frame in flight
 * (LM_ERR_INVALID otherwise).  After LM_ERR_OVERFLOW from collect, submit that frame again.
 *
 * Frames per launch: a 2k-template bank does not fill the chip, so consecutive streamed frames share their kernel launches —
 * the frame's upload starts at once, its front end / coarse pass / refinement / duplicate removal are launched together with
 * those of the following frames once lm_detector_get_batch() (default 8, LM_FRAME_BATCH, lm_detector_set_batch 1..8) frames
 * with the same threshold and class list are waiting, or when lm_detector_flush or a lm_detector_collect that needs one of them
 * is called — or at once while the GPU has fewer than lm_detector_set_batch_queue (default 2, LM_BATCH_QUEUE; 0 = always wait for
 * a full batch) launched batches still to finish: the GPU never idles waiting for a batch to fill, the first frames of a stream go
 * out alone, and batches grow to the maximum exactly when the GPU is the bottleneck.  Results are unchanged and still come back per frame, in order; lm_timings.batch_frames says how many frames the
 * reported launches served.  set_batch(1) restores one launch set per frame (lowest latency). */
int lm_detector_submit_frame(lm_detector *d, const uint8_t *rgb, const uint16_t *depth, int width, int height,
                             float threshold, const char *const *class_ids, int num_class_ids);
int lm_detector_ingest_buffer(lm_detector *d, int width, int height, uint8_t **rgb, uint16_t **depth);
int lm_detector_flush(lm_detector *d);            /* launch the streamed frames that are waiting for their batch to fill */
int lm_detector_set_batch(lm_detector *d, int frames);
int lm_detector_get_batch(const lm_detector *d);
int lm_detector_set_batch_queue(lm_detector *d, int batches);
/* Streamed frames come back in batches: when lm_detector_collect (sort_unique = 1) has seen the event of a batch, the records of ALL its
 * frames are in pinned memory, and helper threads of the library (LM_HOST_THREADS, default 3, never more than the CPUs the process may use
 * leave free, 0 = none; they make no HIP calls) prepare the Detector::match lists of the batch's later frames — record conversion, canonical
 * sort, unique — while the caller takes the first; collecting those frames then only hands the list over.  The same threads copy slices of a
 * submitted frame into the pinned staging buffer beside the caller.  On by default (LM_ASYNC_COLLECT=0 / set_async_collect(d, 0) turn the
 * list preparation off); without helpers the caller does everything itself.  Results are identical either way. */
int lm_detector_set_async_collect(lm_detector *d, int on);
/* Host-side wall time (seconds, accumulated) the library spent on streamed frames: out8 = {frames, staging copy, H2D enqueue, slot
 * bookkeeping, batch launches, collect: waiting for the GPU, record conversion, canonical sort + unique}; reset != 0 clears it. */
int lm_detector_host_profile(lm_detector *d, double *out8, int reset);
/* 1 when the local refinement (similarityLocal, LL.cpp:1366-1428) of the current bank and frame geometry runs on bit planes
 * (kernel k_local_bits), 0 when on byte planes (k_local): which kernel a profile of the match shows.  Valid after a match. */
int lm_detector_refines_on_bit_planes(const lm_detector *d);
/* Which kernels serve similarity (LL.cpp:1284-1354) and similarityLocal (LL.cpp:1366-1428).  Results never depend on it; tests and
 * measurements use it to run every path against the oracle in one process.
 *   refine: 0 = bit planes (k_local_bits; default, any pyramid with a level below the top), 1 = byte strip planes with tiles (k_local),
 *           2 = byte planes, every candidate on its own (k_local without tiles).
 *   coarse: 0 = bit planes (k_coarse_bits; default, used when the refinement runs on bit planes too), 1 = byte linear memories (k_coarse).
 * Refused with frames in flight.  lm_detector_get_paths reports what the current bank and frame geometry actually use (valid after a
 * match): refine 0 / 1 / 2 and coarse 0 / 1 as above. */
int lm_detector_set_paths(lm_detector *d, int refine, int coarse);
int lm_detector_get_paths(const lm_detector *d, int *refine, int *coarse);
/* The response maps of spread / computeResponseMaps / linearize (LL.cpp:1026-1243) as bit planes straight from the quantised images
 * (default, on = 1: wherever the kernels in use read only bit planes, the byte linear memories are not written at all) or as the
 * reference's byte linear memories first, packed into bit planes by a second kernel (on = 0).  Bit 1 (on = 2, 3): the top level's bit
 * planes stay readable after the match (lm_detector_read_stage kind 5; tests).  The top level has three writers — from pixel tiles (T = 4 or 8,
 * a multiple of 8 cells per row: whole bytes), whole dwords per wave (T * T * cells a multiple of 64), ballots OR-ed into a zeroed stream — and
 * the cheapest the geometry allows is used; bit 2 (on = 4 ...) forces the OR-ing writer, bit 3 (on = 8 ...) rules out the tiles (tests).
 * Results never depend on it. */
int lm_detector_set_direct_bits(lm_detector *d, int on);
int lm_detector_last_timings(const lm_detector *d, lm_timings *t);

/* Test/diagnostic access to the device-resident intermediates of the last front end run (parity
 * tests compare them byte-for-byte with the oracle).  kind: 0 quantised colour, 1 quantised
 * normals (W_l*H_l bytes each), 2 linear memories colour, 3 linear memories normals
 * (8*W_l*H_l bytes each; built on demand when the last match ran on bit planes only), 4 the strip
 * records of a level below the top (colour block, normal block: [label][phase][strip][row] 8-byte
 * records of 32 cells x {response is 1, response is 4}), 5 the pair stream of the top level ({is-1
 * dword, is-4 dword} per 32 bytes of the flat linear memories, both modality blocks with their zero
 * tails) — 4 and 5 only after a match that used them (k_local_bits / k_coarse_bits; a stream the front
 * end wrote directly is cleared by the match unless lm_detector_set_direct_bits(d, 2) was set).
 * Copies min(capacity, size) bytes, returns the full size. */
int64_t lm_detector_read_stage(lm_detector *d, int level, int kind, uint8_t *dst, int64_t capacity);

/* Canonical merge of match lists gathered from several ranks (LL.cpp:1771-1776 semantics, A12).
 * In-place on `m` (n entries); returns the new count. */
size_t lm_merge_matches(lm_match *m, size_t n);

/* Caller-side box NMS of the reference driver (linemod_and_levelup_test.py:34-61), numpy `nms`
 * semantics (+1 pixel convention, IoU > thresh suppresses, score-descending, stable for ties by
 * input order as numpy argsort()[::-1] is NOT guaranteed — ties broken by higher index first).
 * boxes: [n][4] float64 x1,y1,x2,y2; scores [n] float64.  keep: [n] int32 out; returns #kept. */
int lm_nms_boxes(const double *boxes, const double *scores, int n, double thresh, int32_t *keep);
/* Translation NMS over refined poses, numpy `nms_norms` of linemod_ros/detect.py:41-51 (used at :128 with thresh 40.0 on
 * poseRefine translations, score = -residual): score-descending, a later pose survives iff ||t_i - t_j|| > thresh.
 * ts: [n][3] float64.  keep: [n] int32 out; returns #kept. */
int lm_nms_norms(const double *ts, const double *scores, int n, double thresh, int32_t *keep);
/* cv::dnn::NMSBoxes on integer rectangles as linemodLevelup/test.cpp:132-144 calls it (40x40 boxes, score_threshold 0,
 * nms_threshold 0.4): the published algorithm of OpenCV 3.4 dnn (NMSFast_) — OpenCV is un-vendored and unpinned, so
 * parity is with that algorithm, not with a binary.  rects: [n][4] int32 x, y, width, height; eta = 1, top_k = 0 are
 * OpenCV's defaults.  keep: [n] int32 out; returns #kept. */
int lm_nms_boxes_cv(const int32_t *rects, const float *scores, int n, float score_threshold, float nms_threshold, float eta,
                    int top_k, int32_t *keep);

void lm_free(void *p);

/* ---- poseRefine (pybind11.cpp:9-14, LL.cpp:27-170) ------------------------------------------
 * scene/model depth: uint16 [height][width] mm; K matrices float32 row-major 3x3; R float32 3x3,
 * t float32 3 (mm).  Outputs: R_out float64 3x3, t_out float64 3 (mm), residual (= ICP fitness,
 * or -1 when the detection window leaves the frame, LL.cpp:52-55).
 * flags: bit0 = LM_ICP_SCENE_FROM_SCENE: register against the SCENE cloud (evident intent) instead
 * of reproducing LL.cpp:109, which down-samples the model cloud twice. */
#define LM_ICP_SCENE_FROM_SCENE 1
typedef struct lm_pose_result {
    double R[9];
    double t[3];
    float residual;      /* fitness; -1 = rejected */
    float inlier_rmse;
    int32_t iterations;
    int32_t n_source;    /* points after voxel down-sampling */
    int32_t n_target;
    int32_t reserved;
} lm_pose_result;

int lm_pose_refine(int device, const uint16_t *scene_depth, const uint16_t *model_depth, int width, int height,
                   const float *scene_K, const float *model_K, const float *model_R, const float *model_t,
                   int detect_x, int detect_y, int flags, lm_pose_result *result);

/* Batched form (top-K hypotheses of one frame, BASELINE config 3): `count` model depth images /
 * poses against one scene depth; cloud preparation, normals and all ICP iterations on the device,
 * one workgroup per hypothesis in the iteration kernel.  device_ms: HIP-event time of the device pipeline. */
int lm_pose_refine_batch(int device, const uint16_t *scene_depth, int width, int height, const float *scene_K,
                         int count, const uint16_t *const *model_depths, const float *model_Ks /*[count][9]*/,
                         const float *model_Rs /*[count][9]*/, const float *model_ts /*[count][3]*/,
                         const int32_t *detect_xy /*[count][2]*/, int flags, lm_pose_result *results,
                         float *device_ms /* may be NULL: HIP-event time of the ICP launches */);

/* ---- resident ICP context (batched poseRefine without per-call allocation) ------
 * The same computation as lm_pose_refine_batch (which is set_scene + set_models + run on a shared
 * per-device context), split so that depth images stay resident in HBM across calls: a frame's scene
 * depth is uploaded once, model depth renderings live in numbered slots and any number of
 * hypotheses may refer to a slot.  Everything of poseRefine::process after the argument checks
 * (LL.cpp:43-148) runs on the device in one stream without host round trips. */
typedef struct lm_icp lm_icp;
int lm_icp_create(int device, lm_icp **out);
void lm_icp_destroy(lm_icp *c);
/* sceneDepth + sceneK of poseRefine::process (LL.cpp:27); defines the frame geometry (changing it drops the slots). */
int lm_icp_set_scene(lm_icp *c, const uint16_t *scene_depth, int width, int height, const float *scene_K);
/* modelDepth images (LL.cpp:27) into slots [first_slot, first_slot + count). */
int lm_icp_set_models(lm_icp *c, int first_slot, int count, const uint16_t *const *model_depths);
/* `count` hypotheses; model_slots == NULL means hypothesis i uses slot i.  Outputs as lm_pose_refine_batch. */
int lm_icp_run(lm_icp *c, int count, const int32_t *model_slots, const float *model_Ks /*[count][9]*/,
               const float *model_Rs /*[count][9]*/, const float *model_ts /*[count][3]*/,
               const int32_t *detect_xy /*[count][2]*/, int flags, lm_pose_result *results, float *device_ms);
/* Test/diagnostic read-back of the last run's device intermediates of one hypothesis.  kind: 0 source
 * cloud, 1 target cloud, 2 target normals (xyz triples, voxel order), 3 {init_guess t[3], T[16],
 * n_model, n_scene, grid_x, grid_y, cell, iterations, 4 phase cycle counts of the iteration kernel}, 4 the slices' partial sums of
 * the last two evaluations [2][64][32] (slots 29-31 of a slice: its shader cycles, search cycles, queued points), 5 the cumulants of the
 * k nearest neighbours per sorted target position [n_target][12].  Copies min(capacity, size) doubles, returns the size. */
int64_t lm_icp_read_debug(lm_icp *c, int hypothesis, int kind, double *dst, int64_t capacity);

/* ---- per-frame pipeline --------------------------------------------------------
 * The loop of the reference driver, linemod_and_levelup_test.py:324-372, as one stream of device work:
 * Detector::match -> dets (x, y, x+width, y+height, similarity) of the matched templates (:331-339) ->
 * numpy nms(dets, 0.5) (:340, :34-61) -> for the first top_k kept matches poseRefine.process against
 * the depth rendering of the matched template view (:349-367).  The renderings (what the driver gets
 * from pysixd's renderer with aTemplateInfo[template_id]'s cam_K / cam_R_w2c / cam_t_w2c) are uploaded
 * once per template and stay in HBM; NMS, top-K, hypothesis set-up and ICP run on the device with no
 * host round trip.  Result = lm_detector_match + lm_nms_boxes + lm_pose_refine_batch on the same frame. */
typedef struct lm_pipeline lm_pipeline;
typedef struct lm_detection {
    lm_match match;          /* the kept match */
    int32_t width, height;   /* its NMS box: the template's size */
    int32_t status;          /* 0 refined; 1 detection window leaves the frame (residual -1, LL.cpp:52-55); 5 no view for this template */
    int32_t reserved;
    lm_pose_result pose;     /* refined pose (status 0) */
} lm_detection;
typedef struct lm_pipeline_timings {
    float match_ms, nms_ms, icp_ms, total_ms;   /* HIP events on the detector's stream */
    int64_t coarse_candidates, matches_pre_unique;
    int32_t icp_iterations, reserved;
} lm_pipeline_timings;
int lm_pipeline_create(lm_detector *det, int width, int height, lm_pipeline **out);
void lm_pipeline_destroy(lm_pipeline *p);
/* Views of templates [first_template, first_template + count) of a class: depth rendering (uint16 [height][width],
 * mm), cam_K, cam_R_w2c (float32 3x3 row-major), cam_t_w2c (float32 3, mm); box_wh: NULL, or [count][2] int32
 * aTemplateInfo 'width','height' (the NMS box of the driver, :335-338); NULL / negative = the template's own size. */
int lm_pipeline_set_views(lm_pipeline *p, const char *class_id, int first_template, int count,
                          const uint16_t *const *depth_ren, const float *Ks, const float *Rs, const float *ts,
                          const int32_t *box_wh);
/* Runs on the detector's resident frame (lm_detector_set_frame / select_frame).  out: top_k entries; *n_out kept. */
/* lm_icp_read_debug on the pipeline's own ICP context (the hypotheses of the last lm_pipeline_run). */
int64_t lm_pipeline_read_icp_debug(lm_pipeline *p, int hypothesis, int kind, double *dst, int64_t capacity);
int lm_pipeline_run(lm_pipeline *p, float threshold, const char *const *class_ids, int num_class_ids, const float *scene_K,
                    int top_k, double nms_iou, int flags, lm_detection *out, int *n_out, lm_pipeline_timings *tm);

/* ---- meshes and rendering --------------------------------------------------------
 * What the reference driver gets from pysixd: inout.load_ply (inout.py) and renderer.render
 * (renderer.py:306-420; linemod_and_levelup_test.py:206-215 for the training views, :352 for the depth_ren
 * poseRefine registers against).  The mesh lives in HBM; a batch of views is rasterised on the device
 * (depth at the native resolution; colour with per-fragment phong shading, light at the eye, at ssaa x the
 * resolution and box-averaged).  OpenCV camera convention: p_cam = R v + t, pixel (i, j) sampled at (i, j). */
typedef struct lm_mesh lm_mesh;
int lm_mesh_create(int device, const float *vertices /*[nv][3] mm*/, const float *normals /*[nv][3] or NULL*/,
                   const uint8_t *colors /*[nv][3] or NULL*/, int nv, const int32_t *faces /*[nf][3]*/, int nf, lm_mesh **out);
int lm_mesh_load_ply(int device, const char *path, lm_mesh **out);   /* ascii / binary_little_endian, triangles */
void lm_mesh_destroy(lm_mesh *m);
int lm_mesh_counts(const lm_mesh *m, int *nv, int *nf);
/* render(model, (width, height), K, R, t, clip_near, clip_far, ambient_weight, shading='phong') for `count` views.
 * Ks/Rs [count][9] float32 row-major, ts [count][3] (mm).  depth_out: uint16 [count][height][width] (mm, truncated
 * like depth.astype(np.uint16)) or NULL; rgb_out: uint8 [count][height][width][3] or NULL. */
int lm_mesh_render(lm_mesh *m, int count, int width, int height, const float *Ks, const float *Rs, const float *ts,
                   float clip_near, float clip_far, float ambient, int ssaa, uint16_t *depth_out, uint8_t *rgb_out);
/* The render_train loop of the driver (:170-252) without the round trip through host images: renders every view
 * (depth + colour), runs Detector::addTemplate on it (mask = depth > 0) and reports the template id per view (-1 where
 * no template could be extracted) plus the extent of the rendered depth (aTemplateInfo 'width','height', :229-236). */
int lm_detector_add_templates_rendered(lm_detector *d, lm_mesh *m, const char *class_id, int count, int width, int height,
                                       const float *Ks, const float *Rs, const float *ts, float clip_near, float clip_far,
                                       float ambient, int ssaa, int32_t *template_ids /*[count]*/, int32_t *box_wh /*[count][2] or NULL*/);
/* Views of a pipeline rendered on the device: depth_ren of template first_template + i = the mesh at (Rs[i], ts[i]). */
int lm_pipeline_set_views_rendered(lm_pipeline *p, lm_mesh *m, const char *class_id, int first_template, int count,
                                   const float *Ks, const float *Rs, const float *ts, float clip_near, float clip_far,
                                   const int32_t *box_wh);

#ifdef __cplusplus
}
#endif
#endif /* AMD_LINEMOD_H */
