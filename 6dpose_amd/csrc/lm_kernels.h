// Internal interface between the host orchestration (detector.cpp, pose_refine.cpp) and the HIP
// kernels (frontend.hip, match.hip, icp.hip).  gfx950 only.  Not part of the public C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lm {

// ---- front end (frontend.hip): reference A1-A7, LL.cpp:350-505, 557-581, 729-880, 1026-1243 ----
void upload_normal_lut(const uint8_t lut400[400]);
// the colour chain (blur7 + sobel_quant + hysteresis) and the normal chain (normals + median5) as single tiled launches
void launch_color_quant(const uint8_t* rgb, float* mag, uint8_t* onehot, int W, int H, float thr_sq, hipStream_t s);
void launch_normals_fused(const uint16_t* depth, uint8_t* raw, uint8_t* med, int W, int H, int dist_thr, int diff_thr, hipStream_t s);
void launch_pyrdown_rgb(const uint8_t* src, uint8_t* dst, int W, int H, hipStream_t s);  // dst (W/2,H/2)
void launch_nn_down2(const uint8_t* src, uint8_t* dst, int W, int H, hipStream_t s);     // dst (W/2,H/2)
// spread (T x T OR) -> 8 response maps -> linearised layout LM[8][T*T][(W/T)*(H/T)] for both modalities of a level in one
// launch ([0] colour, [1] normals); mask[m] may be null; `strips[m]` (may be null) receives the strip-major copy used by
// the refinement kernel.
void launch_build_lm(const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* const lm[2], uint8_t* const strips[2],
                     int W, int H, int T, hipStream_t s);

// several independent jobs of the front end in one launch (frontend.hip, k_fe_stage)
struct LmJob { const uint8_t* quant; const uint8_t* mask; uint8_t* lm; uint8_t* strips; };
// (the bit-plane jobs reuse the two output slots — 24 jobs of a batch of 8 frames must fit the 4 KB of kernel arguments —: `lm` = the strip
// records of the level / the top level's pair stream, `strips` = for the pair stream the flat position of the modality's block in it, as an integer)
enum { kFeNone = 0, kFeColour, kFeNormals, kFePyrDown, kFeNnDown, kFeBuildLm, kFeBitsRows, kFeTopBits, kFeTopBitsAligned, kFeTopBitsTile, kFeBitsRowsTile };
struct FeJob {
    int kind, gx, gy, gz, first;          // job kind, its block grid, its first flat block index (set by launch_fe_stage)
    const void* in; void* out0; void* out1;
    int W, H, a, b;                       // a, b: normals thresholds / pyrDown output size / T
    float f;                              // colour: weak threshold squared
    // Divisions by run-time values cost ~25 instructions each on the GPU and the bodies are short: the quotients a block / thread needs
    // (flat block index -> job-local (bx, by, bz); linear-memory index -> (row, column); phase -> (row, column) inside the T x T cell)
    // come from multipliers the host prepares: n / d = umulhi(n, m), m = ceil(2^32 / d), exact while n * d < 2^32 (m = 0 stands for d = 1).
    uint32_t m_gx, m_gxgy, m_wd, m_t, m_np;  // (m_np: positions of a plane, the aligned pair-stream writer)
    int Wd, Hd;                           // build_lm: decimated size
    LmJob lm[2];                          // build_lm: [0] colour, [1] normals
};
constexpr int kFeMaxJobs = 24;            // 3-4 jobs per frame of a batch (stage 1), kMaxLevels per frame in the last stage: the struct is a kernel argument (< 4 KB)
struct FeStage { int njobs; FeJob job[kFeMaxJobs]; };
static_assert(sizeof(FeStage) + 16 <= 4096, "FeStage is passed by value: kernel arguments are limited to 4 KB");
void fe_job_colour(FeJob& j, const uint8_t* rgb, float* mag, uint8_t* onehot, int W, int H, float thr_sq);
void fe_job_normals(FeJob& j, const uint16_t* depth, uint8_t* raw, uint8_t* med, int W, int H, int dist_thr, int diff_thr);
void fe_job_pyrdown(FeJob& j, const uint8_t* src, uint8_t* dst, int W, int H);
void fe_job_nn_down2(FeJob& j, const uint8_t* src, uint8_t* dst, int W, int H);
void fe_job_build_lm(FeJob& j, const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* const lm[2], uint8_t* const strips[2],
                     int W, int H, int T);
// the bit planes straight from the quantised maps, when nothing reads the byte planes (frontend.hip)
bool fe_bits_rows_possible(int W, int T);     // the level's rows fit the stage's LDS
void fe_job_bits_rows(FeJob& j, const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* const bits[2], int W, int H, int T, bool tiles);   // tiles: from pixel tiles where the geometry allows (T = 4, 5 or 8, the row fits the LDS pool)
// (the writer of whole dwords when the label planes start on 64-position boundaries — fe_top_bits_kind —, else, or when forced, the one that ORs
// shifted ballots into a stream that must be zero beforehand)
// mode: 0 = the cheapest writer the geometry allows (pixel tiles -> whole dwords per wave -> OR-ed ballots), 1 = the OR-ing writer, 2 = no tiles (tests)
void fe_job_top_bits(FeJob& j, const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* stream, const uint32_t bit0[2], int W, int H, int T, int mode);
int fe_top_bits_kind(int W, int H, int T, const uint32_t bit0[2], int mode);   // kFeTopBitsTile / kFeTopBitsAligned / kFeTopBits (the only one that needs a zeroed stream)
void launch_fe_stage(FeStage& st, hipStream_t s);
void launch_fe_bits(FeStage& st, hipStream_t s);      // a launch of bit-plane jobs only (fe_job_bits_rows, fe_job_top_bits)

// ---- matching (match.hip): reference A8-A11, LL.cpp:1284-1428, 1788-1941 ----
struct LevelGeom {        // one pyramid level of the current frame
    int W, H, T, Wd, Hd;  // image size, sampling step, decimated size
    uint32_t lm_off[2];   // byte offset of the colour / normal LM block inside the LM arena
    uint32_t sm_off[2];   // byte offset of the strip-major copy inside the strip arena (levels below the top)
    int NS;               // 16-column strips per plane row = ceil(Wd / 16)
};
constexpr int kMaxLevels = 8;
struct FrameGeom {
    int levels;
    LevelGeom lv[kMaxLevels];
};
constexpr int kFeatBatch = 8;   // features per unrolled batch (independent gathers in flight per lane)
struct TemplEntry {       // one (pyramid, level): both modalities, colour features first
    uint32_t feat_start;  // index into feat_off / feat_xy (multiple of kFeatBatch)
    uint16_t nf;          // true feature count of both modalities (the score denominator)
    uint16_t nf_padded;   // nf rounded up to a multiple of kFeatBatch; padding reads the zero tail
    int32_t width, height;
    int16_t min_x, min_y, max_x, max_y;   // bounding box of the features (fast-path test of the refinement)
};
struct Candidate {        // coarse hit, and (same layout) final match record
    int32_t x, y;
    float score;
    int32_t work;         // index into the work list (-> class position, template id)
};

// Tile refinement (match.hip): candidates of one template whose coarse cells are neighbours share most of their level-0
// windows; k_coarse groups them into tiles, k_local accumulates a tile's window region once for all its members.
constexpr int kTileStep = 4;      // most level-0 cells between the windows of neighbouring coarse cells: 2 * T_top / T_0 <= this
constexpr int kTileMaskBits = 10; // member bits of TileRec::mask (kTileNx x kTileNy); above them the window steps, 3 bits each: columns 1..4, second row
constexpr int kTileNx = 5;        // coarse cells per tile: 16 + 4 * 4 = 32 columns = the part of 3 strips that survives any byte phase
constexpr int kTileNy = 2;        //                        16 + 4 rows; 20 rows x 3 strips = 60 of the 64 lanes of a load
struct TileRec {
    int32_t work;                 // work-list index (-> template pyramid)
    uint32_t gxy;                 // int16 gx0 | int16 gy0 << 16: level-0 cell of the first member's window origin (x / T - 8, LL.cpp:1380)
    uint32_t slot_base;           // its members own the candidate slots [slot_base, slot_base + popcount(mask)), row-major by (j, i)
    uint32_t mask;                // bit i + kTileNx * j: the candidate of coarse cell (c0 + i, r0 + j) is a member; bits kTileMaskBits..: window steps
};
// counters[0] of a frame's working counters: candidates in the low kCandBits bits, tiles planned above them (one atomic of
// k_coarse reserves both)
constexpr int kCandBits = 40;
constexpr unsigned long long kCandMask = (1ull << kCandBits) - 1ull;
// ---- frames per launch --------------------------------------------------------------------------------------------------
// A 2k-template shard does not fill the chip: k_coarse runs at twice, k_local at 1.3x the per-unit cost they reach on a 16k bank
// (launch / drain latency, one partial wave of workgroups).  In stream mode the matching kernels therefore serve a BATCH of up
// to kMaxBatch consecutive frames per launch: every frame of the batch owns its arenas, candidate / record buffers, counters and
// hash table (the per-slot buffers of detector.cpp), the kernels index them by blockIdx.y (k_coarse, k_dedupe) or through the
// flat work-item index (k_local).  A lone frame is a batch of one.
constexpr int kMaxBatch = 8;
constexpr int kCounterWords = 64;             // working counters per result slot: [0] candidates | tiles << kCandBits (k_coarse), [1] distinct,
                                              // [2] alive, [3] key overflow, [7] block ticket of k_dedupe, [8 + 2 s], [9 + 2 s]: 16x16 evaluations /
                                              // their algorithmic bytes of the refinement, shard s = block & (kStatShards - 1)
constexpr int kStatShards = 16;
struct FrameSlot {                            // device pointers of ONE frame of a batch
    const uint8_t* lm_arena;                  // linear memories (flat) of the frame
    const uint8_t* sm_arena;                  // strip-major copy (levels below the top)
    Candidate* cands;                         // [cand_cap] coarse candidates
    TileRec* tiles;                           // [tile_cap] or null
    uint8_t* todo;                            // [cand_cap] or null
    unsigned long long* counters;             // [kCounterWords] working counters, zero between frames
    Candidate* matches;                       // unused (round 2: a pinned host copy of matches_dev written by k_local; the host now fetches matches_dev when asked)
    Candidate* matches_dev;                   // [cap] refined records in HBM, one per candidate (duplicate removal, on-device NMS, exchange)
    unsigned long long* dedupe_table;         // open-addressing table of k_dedupe
    Candidate* distinct;                      // [cap] records without exact duplicates, pinned host memory
    ulonglong2* distinct_keys;                // the same as 128-bit exchange keys (HBM), may be null
    unsigned long long* final_dev;            // [8] published counts of the finished frame (HBM): candidates, distinct, alive, key overflow
    unsigned long long* final_host;           // [8] the same in pinned host memory + [4] tiles, [5] evaluations, [6] algorithmic bytes
};
struct FrameBatch {
    int nb;                                   // frames in this launch (1 .. kMaxBatch)
    int pad;
    FrameSlot f[kMaxBatch];
};

// Bit planes (match.hip).  Levels below the top: per frame of a batch the strip arena the records are packed from
// (launch_pack_bits; the front end writes them itself when nothing reads the strip bytes) and its bit arena — the strip arena's layout at
// half the offsets: per plane row and strip an 8-byte record of 32 cells x {is 1, is 4} instead of a 16-byte row.
struct BitsBatch { const uint8_t* strips[kMaxBatch]; uint8_t* bits[kMaxBatch];
                   uint8_t* top_clear[kMaxBatch]; uint32_t top_clear_units; };   // != 0: k_local_bits zeroes 16 x units bytes of every frame's pair stream (the front end ORs the next frame into it)
constexpr int kBitsSmallMax = 511;            // features per template entry the 9-bit counters hold; larger entries (<= 16383) take the 14-bit instantiation
void launch_pack_bits(const BitsBatch& B, int nb, const LevelGeom& lv, hipStream_t s);
// Every level below the top: todo[ci] = 1 for the candidates it leaves to launch_local's per-candidate path (windows leaving their planes);
// max_features = the largest nf of the bank's entries below the top level.
void launch_local_bits(const FrameBatch& fb, const BitsBatch& B, const FrameGeom& g, const TemplEntry* entries, const uint32_t* feat_word,
                       const int32_t* work_pyramids, uint32_t cand_cap, float threshold, uint32_t cap, uint32_t dedupe_cap_slots, int grid_blocks,
                       int max_features, hipStream_t s);
// Coarse pass on bit planes: per frame of a batch the flat arena and the pair stream of bytes [byte0, byte0 + 32 npairs) of it (the top
// level's blocks of both modalities with their zero tails): launch_pack_top packs it from the bytes, the front end writes it directly when
// nothing reads the top level's bytes (frontend.hip, top_bits_body: the stream must be zero before).
struct TopBits { const uint8_t* lm[kMaxBatch]; uint8_t* bits[kMaxBatch]; };
void launch_pack_top(const TopBits& B, int nb, uint32_t byte0, uint32_t npairs, hipStream_t s);
// candidates only (no tiles); max_features = the largest nf of the bank's top-level entries
void launch_coarse_bits(const FrameBatch& fb, const TopBits& B, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off,
                        const int32_t* work_pyramids, int num_work, float threshold, uint32_t cap, uint32_t byte0, int max_features, hipStream_t s);
bool tile_plan_possible(const FrameGeom& g);
size_t coarse_plan_lds_bytes(int Wd, int Hd);
// Per frame of the batch: counters[0] = number of candidates produced (may exceed cap: nothing is written past cap); with tiles
// (and a geometry tile_plan_possible accepts) also counters[0] >> kCandBits = number of tiles, todo[slot] = 1 for the candidates
// no tile serves.  Grid (workgroups of templates, frames).
void launch_coarse(const FrameBatch& fb, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off,
                   const int32_t* work_pyramids, int num_work, float threshold, uint32_t cap, uint32_t tile_cap, hipStream_t s);
// Persistent grid: waves stride over the tiles of all frames of the batch, then over their candidates (counts read on the device).
// matches_dev[ci] = refined candidate ci (work = -1: dropped).  Also empties the part of every frame's hash table k_dedupe will use.
// Per feature of a level below the top ONE word, feat_word = base0 | cls: base0 = byte offset (a multiple of 16) inside the strip
// arena of the 16-byte row of the feature's own cell — plane base + ((lx >> 4) * Hd + ly) * 16 — and cls = lx & 15, the alignment
// class (features of an entry are sorted by it).  The window origin of a work item (gx, gy) adds the same K = ((gx >> 4) * Hd + gy)
// * 16 to every feature, plus one strip (Hd * 16) for the classes whose column carries into the next strip (cls + (gx & 15) >= 16):
// both are folded into the lanes' VGPR offset once per class run, and the word (class bits masked) is the load's scalar offset — one
// scalar instruction per feature for the address (round 2: 9, and 8 for the run bookkeeping).
// run_mask[f / 8] bit (f % 8): feature f starts a new class run (class change, or 62 features of one class: the packed-byte sums hold
// 63 x 4).  Runs have even length and start at even indices (the per-candidate path loads same-class pairs).
constexpr int kRunMax = 62;
void launch_local(const FrameBatch& fb, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off, const uint32_t* feat_word,
                  const uint32_t* run_mask, const uint32_t* feat_xy, const int32_t* work_pyramids, uint32_t cand_cap, float threshold, uint32_t cap,
                  uint32_t dedupe_cap_slots, uint32_t tile_cap, int grid_blocks, hipStream_t s);
// slots of k_dedupe's open-addressing table used for n records (power of two, >= 2n, <= cap_slots = dedupe_table_slots(cand_cap)):
// k_local empties exactly these, k_dedupe hashes into exactly these
__host__ __device__ inline uint32_t dedupe_slots_for(uint32_t n, uint32_t cap_slots) {
    uint32_t t = 1024;
    while (t < 2u * n && t < cap_slots) t <<= 1;
    return t < cap_slots ? t : cap_slots;
}

// ---- on-device NMS + top-K (nms.hip): the caller-side loop of linemod_and_levelup_test.py:331-352 ----
struct TopkSel {          // one kept detection
    int32_t x, y;
    float similarity;
    int32_t work;         // work-list index
    int32_t class_index, template_id;
    int32_t width, height;   // level-0 template size (the NMS box)
};
// Greedy NMS (numpy `nms` of the driver, IoU with the +1 pixel convention, suppress when IoU > thresh) over
// the canonical match list of the frame (sorted + adjacent-unique as Detector::match returns it), stopped
// after top_k kept boxes.  matches_dev / counters as written by launch_local; scratch: topk_nms_scratch_bytes(cap).
// sel[0..*nsel) in keep order; status: 0 ok, 1 field overflow (template id >= 2^24, class >= 128, |x|,|y| >= 2^15).
void launch_topk_nms(const Candidate* matches_dev, const unsigned long long* counters, uint32_t cap, const int32_t* work_pyramids,
                     const int32_t* work_cls, const int32_t* work_tid, const TemplEntry* entries, int levels,
                     const int32_t* class_base /*class position -> first view slot, may be null*/,
                     const int32_t* view_wh /*[views][2] box override, -1 = template size, may be null*/, int num_views, int top_k,
                     double iou_thresh, void* scratch, TopkSel* sel, int32_t* nsel_status, hipStream_t s);
size_t topk_nms_scratch_bytes(uint32_t cap);
// Exact-duplicate removal of the refined records of every frame of a batch (nms.hip), grid (blocks, frames): counters[1] / [2]
// receive the distinct / alive counts; `distinct` (pinned host memory) the surviving records, unordered.  The last block of a frame
// to finish publishes its counts — final_dev[0..3] (HBM: on-device NMS, exchange) and final_host[0..6] (pinned) — and zeroes the
// frame's working counters for the slot's next frame, so that no memset / copy node surrounds the matching kernels.
void launch_dedupe(const FrameBatch& fb, uint32_t cap, size_t table_slots, const int32_t* work_cls, const int32_t* work_tid, int blocks,
                   hipStream_t s);
size_t dedupe_table_slots(uint32_t cap);

// ---- template extraction on the device (train.hip; LL.cpp:589-643, 888-966, 279-318) ----
constexpr int kTrainMaxFeatures = 1024;       // features per template the selection kernel keeps in LDS
constexpr uint32_t kTrainCap = 16384;         // candidates per (view, level, modality) sorted in LDS (128 KB); more -> host path
struct TrainGeom {                            // the maps of the view being prepared (the detector's per-level buffers + scratch)
    int levels;
    int W[kMaxLevels], H[kMaxLevels];
    const float* mag[kMaxLevels];
    const uint8_t* ang[kMaxLevels];
    const uint8_t* nrm[kMaxLevels];
    uint8_t* mask[kMaxLevels];                // object mask pyramid
    uint8_t* lab[kMaxLevels];                 // normal label + 1 inside the twice-eroded mask, else 0
    int32_t* hrun[kMaxLevels];                // distance to the end of the pixel's same-label run along its row
};
// keys_view: [levels][2][cap] sort keys / (distance, position, label) records; counts_view: [levels][16]; bbox_view: [4] = {max(-x), max(-y), max(x), max(y)}
// user_mask (W0 x H0, nonzero = object) replaces depth > 0 when given
void launch_train_prep(const uint16_t* depth, const uint8_t* user_mask, const TrainGeom& g, float strong_sq, int extract_threshold, unsigned long long* keys_view,
                       uint32_t cap, uint32_t* counts_view, int32_t* bbox_view, hipStream_t s);
// out: [views][levels][2][4 + 3 * nf_cap]: status (1 ok, 0 too few candidates, 2 leave it to the host path), count, -, -, then x, y, label
int launch_train_select(const unsigned long long* keys, const uint32_t* counts, const TrainGeom& g, uint32_t cap, int num_features, int nf_cap,
                        int views, int32_t* out, hipStream_t s);

// ---- multi-GPU exchange of match records (exchange.hip) ----
// Block a rank contributes to the all-gather: 4 header words {count, flags, capacity, 0} + capacity 128-bit keys (a sorted run).
constexpr uint32_t kXchgRunOverflow = 1;      // more distinct records than the block holds (count says how many)
constexpr uint32_t kXchgCandOverflow = 2;     // the candidate buffer of the matching kernels overflowed: the frame has to be rerun
constexpr uint32_t kXchgFieldOverflow = 4;    // |x|,|y| >= 32768: not representable in the key
constexpr int kXchgHeaderWords = 256;         // result: {records, flags, world, capacity, -, -, -, -, count per rank...}, then 5-word records
constexpr uint32_t kXchgMaxCapacity = 65536;   // per rank and frame; the ranking kernels take any power of two (more tiles / passes), the blocks and the
                                               // copy-out grow with it, so callers start small (4096) and double on demand
// 128-bit sort key of a match: ascending key order = canonical order of SURVEY A12 (similarity desc, template id, class position, y, x);
// it holds every field, so the records travel as keys.  .x = ~orderable(similarity) << 32 | template id, .y = class << 32 | y+32768 << 16 | x+32768.
__device__ __forceinline__ ulonglong2 xchg_make_key(int x, int y, float sim, int cls, int tid) {
    uint32_t u = __float_as_uint(sim);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;          // orders like the float
    return make_ulonglong2(((unsigned long long)(~u) << 32) | (uint32_t)tid,
                           ((unsigned long long)(uint32_t)cls << 32) | ((unsigned long long)(uint16_t)(y + 32768) << 16) | (uint16_t)(x + 32768));
}
__device__ __forceinline__ bool xchg_key_fits(int x, int y, int cls, int tid) { return x >= -32768 && x <= 32767 && y >= -32768 && y <= 32767 && cls >= 0 && tid >= 0; }
// counters: [0] coarse candidates, [1] distinct records, [3] != 0: a record did not fit the key (all written by the matching stream)
// The exchange kernels serve a GROUP of frames per launch (grid.y / grid.z = frame): per frame the distinct keys and published counters of
// its result slot, a scratch for the sorted 256-key runs, the block it contributes, and — after the all-gather — this frame's block of rank 0
// (rank j's lies j * rank stride further) and the merged list.
struct XchgFrame {
    const ulonglong2* keys; const unsigned long long* counters; ulonglong2* runs; uint32_t* block;     // pack
    const uint32_t* recv; int32_t* merged;                                                               // merge
    int32_t* host;                                                                                       // the merged list's pinned host copy (device pointer), may be null
};
struct XchgGroup { int n; int pad; XchgFrame f[kMaxBatch]; };
// counters: [0] coarse candidates, [1] distinct records, [3] != 0: a record did not fit the key (all written by the matching stream)
void launch_exchange_pack_group(const XchgGroup& G, uint32_t cand_cap, uint32_t cap, hipStream_t s);
// rank_stride_words: distance between consecutive ranks' blocks (0 = packed: one block per rank; a gathered group of frames has the blocks of
// all its frames between two ranks)
void launch_exchange_merge_group(const XchgGroup& G, int world, uint32_t cap, hipStream_t s, uint32_t rank_stride_words = 0);

}  // namespace lm
