// Internal interface between pose_refine.cpp (host) and icp.hip (kernels).  gfx950 only.
// The whole of poseRefine::process (LL.cpp:27-155) after the argument checks runs on the device:
// bounding box + dilated mask + back-projection + centroid init (LL.cpp:43-104), the two
// VoxelDownSample calls (LL.cpp:108-109), EstimateNormals (LL.cpp:127) and RegistrationICP
// (LL.cpp:128-130).  All buffers are sized for the worst case (every pixel of the frame a point):
// 288 GB of HBM makes that free and removes every host round trip between the stages.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lm {

constexpr int kIcpGrid = 64;                      // NN search grid: at most 64 x 64 ...
constexpr int kIcpCells = kIcpGrid * kIcpGrid + 1; // ... columns in x and y (+1 end marker)
constexpr int kIcpCells16 = 4104;                 // kIcpCells rounded up to a multiple of 8 (16-byte copies of the u16 table)
constexpr int kIcpStrips = 48;                    // row strips of the bounding box in k_icp_points
constexpr int kIcpMaxSplit = 64;                  // workgroups (source slices) per hypothesis in k_icp_search
constexpr int kIcpSortGroups = 8;                 // workgroups that sort a cloud, each a contiguous range of the leading key coordinate (k_icp_voxel_wide / k_icp_grid_wide)
constexpr int kIcpCovStride = 12;                 // 9 cumulants, neighbour count, squared nearest-neighbour separation, pad

struct __attribute__((aligned(16))) TgtRec { double x, y, z; int orig; int zq; };    // target point as staged in LDS (32 B): xyz, original index, quantised depth

struct IcpIn {               // one pose hypothesis (uploaded)
    float mK[9];             // model camera matrix (row-major 3x3, float like the reference's cv::Mat_<float>)
    int dx, dy;              // detectX, detectY
    int model_slot;          // which resident model depth image
    int pad;
};

enum IcpStatus {             // IcpState::status; every value has ONE meaning (the host tests them by name)
    kIcpOk = 0,
    kIcpOutOfFrame = 1,      // window leaves the frame (LL.cpp:52-55): residual = -1
    kIcpEmptyModel = 2,      // empty model depth
    kIcpTooLarge = 3,        // cloud too large for 64-bit voxel keys
    kIcpNoDetection = 4,     // pipeline: hypothesis slot without a detection (k_icp_bind)
    kIcpNoView = 5,          // pipeline: the matched template has no rendered view (k_icp_bind)
    kIcpStalled = 6,         // a strip of k_icp_points_fused waited a second for the strips before it (the GPU was taken away from the launch?)
};

struct IcpState {            // one pose hypothesis (device-written, downloaded after the run)
    int bbox[4];             // x0,y0,x1,y1 of modelDepth > 0 (uploaded as INT_MAX,INT_MAX,-1,-1)
    int status;              // IcpStatus below
    int n_model, n_scene;    // back-projected points
    int n_src, n_tgt;        // after voxel down-sampling
    int gx, gy, zq_max;      // search grid: columns in x and y, largest quantised depth
    int iterations, n_corr;
    double init[3];          // init_guess translation (LL.cpp:101-104)
    double gminx, gminy, gminz, cell, inv_cell, inv_z;
    double T[16];            // final transformation_ (row-major)
    double fitness, rmse;    // fitness_, inlier_rmse_
    int stop;                // RegistrationICP finished (converged or max_iteration)
    int n_far;               // target points k_icp_knn left to k_icp_knn_far (their k nearest are more than 8 rings away)
    double fit_hist[2], rmse_hist[2];   // fitness / rmse of the last two evaluations, slot = evaluation parity
    int vox_done[2];         // groups of k_icp_voxel_wide that finished the model / scene cloud (kIcpSortGroups: k_icp_voxel has nothing to do)
    int grid_done;           // the same for k_icp_grid_wide / k_icp_grid
    int team_size;           // k_icp_team: workgroups at work on the hypothesis in the current launch
    int resume_it;           // k_icp_team: > 0 = the hypothesis was suspended after the finish stage of this evaluation index (T, fit_hist, rmse_hist hold what the next launch goes on from)
    int team_note[4];        // k_icp_team left the hypothesis to the sliced launches: reason (1 source slice too large, 2 grid, 3 slab overflow: + member, targets needed, capacity; 4 time-out), else 0
    long long vox_clk[4];    // k_icp_voxel diagnostics (model cloud): cycles for the extent, the keys, the sort, the voxel means
    long long sort_clk[16];  // k_icp_voxel_wide (0-7, model cloud) / k_icp_grid_wide (8-15) diagnostics, slowest group per phase: cycles for picking its points, the sort, (voxels: count + wait for the groups before), writing, (grid: the column table); 6 / 13: largest group
    long long knn_clk[4];    // k_icp_knn diagnostics: slowest workgroup's cycles staging, in the 8-lane trips, in the whole-wave pass; points handed to whole waves
    long long clk[8];        // k_icp_loop shader cycles (thread 0): A1 certainty test, reduction, solve, transform, A2 search, accumulate, queued points, -
};

struct IcpBuffers {
    const uint16_t* scene;   // [H][W]
    const uint16_t* models;  // [slots][H][W]
    int* model_bbox;         // [slots][8] box of modelDepth > 0 of a resident image as INT_MAX - x0, INT_MAX - y0, x1 + 1, y1 + 1 (all zero = empty) and a state word (1: known; worked out at upload by k_icp_model_boxes, or by the first run that uses the slot)
    const IcpIn* in;         // [count]
    IcpState* st;            // [count]
    float sK[9];             // scene camera matrix
    int count;               // hypotheses of this run
    size_t cap;              // points per hypothesis and cloud (= W*H)
    size_t cap2;             // cap rounded up to a power of two (sort scratch)
    double* model_pts;       // [count][cap][3]
    double* scene_pts;       // [count][cap][3]
    double* src;             // [count][cap][3]  VoxelDownSample(model)
    double* tgt;             // [count][cap][3]  VoxelDownSample(scene) (scene-from-scene mode only)
    double* tgt_sorted;      // [count][cap][3]  target points in grid-cell order
    int* tgt_orig;           // [count][cap]     original index of the point at a sorted position
    TgtRec* tgt_rec;         // [count][cap]     the same points as 32-byte records (xyz + original index): LDS staging source
    int* cell_start;         // [count][kIcpCells]
    unsigned short* cell_start16;  // [count][kIcpCells16] the same table in 16 bits (valid when n_tgt < 65536)
    double* cov;             // [count][cap][kIcpCovStride] cumulants of the k nearest neighbours (sorted positions)
    double* normals;         // [count][cap][3]  per sorted position
    double* work;            // [count][cap][3]  transformed source cloud
    int* prev_nn;            // [count][cap]     previous correspondence (sorted position) of every source point
    double* nn_lb;           // [count][cap]     lower bound on the distance to the nearest target of a point without correspondence
    int* strip_cnt;          // [count][kIcpStrips][2] model / scene points per strip
    double* strip_sum;       // [count][kIcpStrips][8] centroid sums per strip
    unsigned long long* strip_pub; // [count][kIcpStrips] k_icp_points_fused: bit 63 = counted, scene points << 32 | model points (cleared by k_icp_bbox)
    double* strip_mm;        // [count][kIcpStrips][12] min xyz, max xyz of the strip's model points, then of its scene points
    unsigned int* sort_look; // [count][2][kIcpSortGroups] k_icp_voxel_wide: voxels of a group + 1 (0 = not yet known; zeroed by k_icp_points)
    double* partial;         // [2][count][kIcpMaxSplit][32] partial sums of one ICP evaluation, double-buffered by evaluation parity
    unsigned long long* keys;// [count][2][cap2] sort scratch for lists longer than the LDS capacity
    unsigned long long* xchg;// [2][count][kIcpMaxSplit][64] the sums the members of a k_icp_team team publish: 8-byte granules {half of a sum, tag}
};

struct TopkSel;
void launch_icp_bind(const TopkSel* sel, const int32_t* nsel_status, const int32_t* class_base, const float* view_K,
                     const int32_t* view_valid, int num_views, IcpIn* in, IcpState* st, int top_k, hipStream_t s);
// solo_from == 0: RegistrationICP as one launch (k_icp_team: a team of workgroups per hypothesis, all evaluations inside); a hypothesis whose
// clouds it cannot hold comes back with stop == 0 and the caller runs launch_icp_evals(0 .. max_iter + 1) for it.  Otherwise: sliced launches only.
void launch_icp_pipeline(const IcpBuffers& B, int count, int W, int H, int flags, double voxel, double max_dist, int max_iter,
                         double rel_tol, int knn, int solo_from, hipStream_t s);
// bit 0x100 of launch_icp_pipeline's flags: every model slot the hypotheses use had its box worked out at upload (launch_icp_model_boxes)
void launch_icp_model_boxes(const uint16_t* models, int* model_bbox, int first_slot, int count, int W, int H, hipStream_t s);
void launch_icp_evals(const IcpBuffers& B, int count, int it_from, int it_to, double max_dist, int max_iter, double rel_tol, hipStream_t s);
// the team kernel alone: large == 0 what launch_icp_pipeline launches at its end; large == 1 the builds for more than 704 source points per
// workgroup, which the caller tries on unfinished hypotheses (stop == 0) before launch_icp_evals
void launch_icp_team(const IcpBuffers& B, int count, int large, double max_dist, int max_iter, double rel_tol, hipStream_t s);

}  // namespace lm
