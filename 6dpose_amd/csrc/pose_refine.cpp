// poseRefine::process (LL.cpp:27-155) behind the C ABI.  The host only validates arguments, stages
// the depth images through pinned memory and composes the final [R|t] from the device result
// (LL.cpp:146-154); everything between — bounding box, dilated mask, back-projection, centroid
// init, both VoxelDownSample calls, EstimateNormals and the whole ICP loop — runs in icp.hip with
// no host round trip.  An `lm_icp` context owns one HIP stream, the resident depth images and the
// worst-case-sized arenas (W*H points per hypothesis and cloud: 288 GB of HBM make that free).
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/amd_linemod.h"
#include "icp_internal.h"

using namespace lm;

int lm_set_error(int code, const char* fmt, ...);   // detector.cpp

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return lm_set_error(LM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                                  __FILE__, __LINE__);                                        \
    } while (0)

namespace {
constexpr double kVoxel = 0.0025;     // LL.cpp:106
constexpr double kMaxDist = 0.01;     // LL.cpp:31
constexpr int kMaxIter = 30;          // open3d ICPConvergenceCriteria default
constexpr double kRelTol = 1e-6;
constexpr int kKnn = 30;              // open3d KDTreeSearchParamKNN default

size_t pow2_at_least(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}
}  // namespace

namespace {

void free_arenas(lm_icp* c) {
    void* ptrs[] = {c->B.model_pts, c->B.scene_pts, c->B.src, c->B.tgt, c->B.tgt_sorted, c->B.tgt_orig, c->B.cell_start,
                    c->B.cov, c->B.normals, c->B.work, c->B.prev_nn, c->B.nn_lb, c->B.partial, c->B.strip_cnt, c->B.strip_sum, c->B.tgt_rec, c->B.cell_start16, c->B.keys, c->B.xchg, c->B.strip_mm, c->B.strip_pub, c->B.sort_look, c->d_in, c->d_st};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->B = IcpBuffers{};
    c->d_in = nullptr; c->d_st = nullptr;
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_st) (void)hipHostFree(c->h_st);
    if (c->h_st2) (void)hipHostFree(c->h_st2);
    c->h_in = nullptr; c->h_st = nullptr; c->h_st2 = nullptr; c->h_cap = 0;
    c->max_count = 0;
}

int ensure_pinned(lm_icp* c, size_t bytes) {
    if (bytes <= c->pinned_bytes) return LM_OK;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinned_bytes = 0;
    HIP_TRY(hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault));
    c->pinned_bytes = bytes;
    return LM_OK;
}

}  // namespace
int lm_icp_ensure_arenas(lm_icp* c, int count) {
    if (count <= c->max_count) return LM_OK;
    HIP_TRY(hipStreamSynchronize(c->s));
    free_arenas(c);
    const int n = std::max(count, 16);
    const size_t cap = (size_t)c->W * c->H, cap2 = pow2_at_least(cap);
    const size_t pts = (size_t)n * cap * 3 * sizeof(double);
    IcpBuffers& B = c->B;
    B.cap = cap; B.cap2 = cap2;
    HIP_TRY(hipMalloc((void**)&B.model_pts, pts));
    HIP_TRY(hipMalloc((void**)&B.scene_pts, pts));
    HIP_TRY(hipMalloc((void**)&B.src, pts));
    HIP_TRY(hipMalloc((void**)&B.tgt, pts));
    HIP_TRY(hipMalloc((void**)&B.tgt_sorted, pts));
    HIP_TRY(hipMalloc((void**)&B.normals, pts));
    HIP_TRY(hipMalloc((void**)&B.work, pts));
    HIP_TRY(hipMalloc((void**)&B.cov, (size_t)n * cap * kIcpCovStride * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&B.tgt_orig, (size_t)n * cap * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&B.prev_nn, (size_t)n * cap * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&B.nn_lb, (size_t)n * cap * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&B.partial, (size_t)2 * n * kIcpMaxSplit * 32 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&B.strip_cnt, (size_t)n * kIcpStrips * 2 * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&B.strip_sum, (size_t)n * kIcpStrips * 8 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&B.strip_pub, (size_t)n * kIcpStrips * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(B.strip_pub, 0, (size_t)n * kIcpStrips * sizeof(unsigned long long)));                      // (0 = not counted yet; every run leaves them cleared for the next)
    HIP_TRY(hipMalloc((void**)&B.strip_mm, (size_t)n * kIcpStrips * 12 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&B.sort_look, (size_t)n * 2 * lm::kIcpSortGroups * sizeof(unsigned int)));
    HIP_TRY(hipMalloc((void**)&B.tgt_rec, (size_t)n * cap * sizeof(TgtRec)));
    HIP_TRY(hipMalloc((void**)&B.cell_start16, (size_t)n * kIcpCells16 * sizeof(unsigned short)));
    HIP_TRY(hipMalloc((void**)&B.cell_start, (size_t)n * kIcpCells * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&B.keys, (size_t)n * 2 * cap2 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void**)&B.xchg, (size_t)2 * n * lm::kIcpMaxSplit * 64 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(B.xchg, 0, (size_t)2 * n * lm::kIcpMaxSplit * 64 * sizeof(unsigned long long)));      // tag 0 = never published (runs count from 1)
    HIP_TRY(hipDeviceSynchronize());                                 // (the ICP stream does not wait for the null stream)
    HIP_TRY(hipMalloc((void**)&c->d_in, (size_t)n * sizeof(IcpIn)));
    HIP_TRY(hipMalloc((void**)&c->d_st, (size_t)n * sizeof(IcpState)));
    HIP_TRY(hipHostMalloc((void**)&c->h_in, (size_t)n * sizeof(IcpIn), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&c->h_st, (size_t)n * sizeof(IcpState), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&c->h_st2, (size_t)n * sizeof(IcpState), hipHostMallocDefault));
    c->h_cap = n;
    c->max_count = n;
    return LM_OK;
}

int lm_icp_set_geometry(lm_icp* c, int W, int H) {
    if (W == c->W && H == c->H) return LM_OK;
    HIP_TRY(hipStreamSynchronize(c->s));
    free_arenas(c);
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->d_models) (void)hipFree(c->d_models);
    if (c->d_model_bbox) (void)hipFree(c->d_model_bbox);
    c->d_scene = nullptr; c->d_models = nullptr; c->d_model_bbox = nullptr; c->slots = 0; c->have_scene = false;
    c->slot_boxed.clear();
    c->W = W; c->H = H;
    HIP_TRY(hipMalloc((void**)&c->d_scene, (size_t)W * H * sizeof(uint16_t)));
    return LM_OK;
}

int lm_icp_ensure_slots(lm_icp* c, int slots) {
    if (slots <= c->slots) return LM_OK;
    const size_t img = (size_t)c->W * c->H * sizeof(uint16_t);
    const int n = std::max(slots, std::max(16, c->slots * 2));
    uint16_t* p = nullptr;
    HIP_TRY(hipMalloc((void**)&p, (size_t)n * img));
    if (c->d_models) {
        HIP_TRY(hipMemcpyAsync(p, c->d_models, (size_t)c->slots * img, hipMemcpyDeviceToDevice, c->s));
        HIP_TRY(hipStreamSynchronize(c->s));
        (void)hipFree(c->d_models);
    }
    c->d_models = p;
    int* boxes = nullptr;
    HIP_TRY(hipMalloc((void**)&boxes, (size_t)n * 8 * sizeof(int)));
    HIP_TRY(hipMemset(boxes, 0, (size_t)n * 8 * sizeof(int)));
    if (c->d_model_bbox) {
        HIP_TRY(hipMemcpy(boxes, c->d_model_bbox, (size_t)c->slots * 8 * sizeof(int), hipMemcpyDeviceToDevice));
        (void)hipFree(c->d_model_bbox);
    }
    HIP_TRY(hipDeviceSynchronize());
    c->d_model_bbox = boxes;
    c->slot_boxed.resize((size_t)n, 0);
    c->slots = n;
    return LM_OK;
}


void lm_icp_compose_result(const IcpState& st, const float* model_R, const float* model_t, lm_pose_result* op) {
    lm_pose_result& o = *op;
    // init_base (float): [R|t], only t.z / 1000 (LL.cpp:34-39)
    float base[16];
    for (int k = 0; k < 16; ++k) base[k] = 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) base[4 * r + q] = model_R[3 * r + q];
        base[4 * r + 3] = model_t[r];
    }
    base[11] = base[11] / 1000.0f;
    base[15] = 1.f;
    // result = transformation_ * init_base.cast<double>() (LL.cpp:146); t * 1000 (LL.cpp:154)
    double M[16];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            double v = 0;
            for (int k = 0; k < 4; ++k) v += st.T[4 * a + k] * (double)base[4 * k + b];
            M[4 * a + b] = v;
        }
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) o.R[3 * a + b] = M[4 * a + b];
        o.t[a] = M[4 * a + 3] * 1000.0;
    }
    o.residual = (float)st.fitness;       // residual = fitness_ (LL.cpp:148)
    o.inlier_rmse = (float)st.rmse;
    o.iterations = st.iterations;
    o.n_source = st.n_src;
    o.n_target = st.n_tgt;
}

bool lm_icp_unfinished(const IcpState* st, int count) {
    for (int i = 0; i < count; ++i)
        if (st[i].status == 0 && st[i].stop == 0) return true;
    return false;
}

extern "C" int lm_icp_create(int device, lm_icp** out) {
    if (!out) return lm_set_error(LM_ERR_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lm_set_error(LM_ERR_NO_DEVICE, "no HIP device visible; libamdlinemod has no CPU fallback");
    if (device < 0 || device >= ndev) return lm_set_error(LM_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    lm_icp* c = new lm_icp();
    c->device = device;
    if (const char* e = getenv("LM_ICP_SLICED")) c->solo_from = e[0] && e[0] != '0' ? -1 : 0;
    if (hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->e0) != hipSuccess ||
        hipEventCreate(&c->e1) != hipSuccess) {
        delete c;
        return lm_set_error(LM_ERR_HIP, "could not create the ICP stream / events");
    }
    *out = c;
    return LM_OK;
}

extern "C" void lm_icp_destroy(lm_icp* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->s) (void)hipStreamSynchronize(c->s);
    free_arenas(c);
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->d_models) (void)hipFree(c->d_models);
    if (c->d_model_bbox) (void)hipFree(c->d_model_bbox);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->e0) (void)hipEventDestroy(c->e0);
    if (c->e1) (void)hipEventDestroy(c->e1);
    if (c->s) (void)hipStreamDestroy(c->s);
    delete c;
}

extern "C" int lm_icp_set_scene(lm_icp* c, const uint16_t* scene_depth, int width, int height, const float* scene_K) {
    if (!c || !scene_depth || !scene_K || width <= 0 || height <= 0) return lm_set_error(LM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = lm_icp_set_geometry(c, width, height);
    if (rc) return rc;
    const size_t img = (size_t)width * height * sizeof(uint16_t);
    HIP_TRY(hipStreamSynchronize(c->s));                            // the staging buffer may still be in flight
    if ((rc = ensure_pinned(c, img))) return rc;
    memcpy(c->pinned, scene_depth, img);
    HIP_TRY(hipMemcpyAsync(c->d_scene, c->pinned, img, hipMemcpyHostToDevice, c->s));
    memcpy(c->sK, scene_K, sizeof(c->sK));
    c->have_scene = true;
    return LM_OK;
}

extern "C" int lm_icp_set_models(lm_icp* c, int first_slot, int count, const uint16_t* const* model_depths) {
    if (!c || first_slot < 0 || count < 0 || (count && !model_depths)) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (c->W <= 0) return lm_set_error(LM_ERR_INVALID, "lm_icp_set_scene must define the frame geometry first");
    if (count == 0) return LM_OK;
    for (int i = 0; i < count; ++i)
        if (!model_depths[i]) return lm_set_error(LM_ERR_INVALID, "model depth %d is null", i);
    HIP_TRY(hipSetDevice(c->device));
    int rc = lm_icp_ensure_slots(c, first_slot + count);
    if (rc) return rc;
    const size_t img = (size_t)c->W * c->H * sizeof(uint16_t);
    HIP_TRY(hipStreamSynchronize(c->s));
    // the scene image may sit at the start of the staging buffer and still be copying: keep it, append after it
    if ((rc = ensure_pinned(c, img * (size_t)(count + 1)))) return rc;
    uint8_t* stage = (uint8_t*)c->pinned + img;
    for (int i = 0; i < count; ++i) memcpy(stage + (size_t)i * img, model_depths[i], img);
    HIP_TRY(hipMemcpyAsync(c->d_models + (size_t)first_slot * c->W * c->H, stage, img * (size_t)count, hipMemcpyHostToDevice, c->s));
    lm::launch_icp_model_boxes(c->d_models, c->d_model_bbox, first_slot, count, c->W, c->H, c->s);   // the boxes of the new images, once (LL.cpp:43-50)
    for (int i = 0; i < count; ++i) c->slot_boxed[(size_t)first_slot + i] = 1;
    return LM_OK;
}

extern "C" int lm_icp_run(lm_icp* c, int count, const int32_t* model_slots, const float* model_Ks, const float* model_Rs,
                          const float* model_ts, const int32_t* detect_xy, int flags, lm_pose_result* results, float* device_ms) {
    if (!c || count < 0 || (count && (!model_Ks || !model_Rs || !model_ts || !detect_xy || !results)))
        return lm_set_error(LM_ERR_INVALID, "null argument");
    if (device_ms) *device_ms = 0.f;
    if (count == 0) return LM_OK;
    if (!c->have_scene) return lm_set_error(LM_ERR_INVALID, "no scene depth resident (lm_icp_set_scene)");
    for (int i = 0; i < count; ++i) {
        const int slot = model_slots ? model_slots[i] : i;
        if (slot < 0 || slot >= c->slots) return lm_set_error(LM_ERR_INVALID, "model slot %d of hypothesis %d is not resident", slot, i);
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = lm_icp_ensure_arenas(c, count);
    if (rc) return rc;
    for (int i = 0; i < count; ++i) {
        IcpIn& in = c->h_in[i];
        memcpy(in.mK, model_Ks + 9 * i, sizeof(in.mK));
        in.dx = detect_xy[2 * i]; in.dy = detect_xy[2 * i + 1];
        in.model_slot = model_slots ? model_slots[i] : i;
        in.pad = 0;
        IcpState& st = c->h_st[i];
        memset(&st, 0, sizeof(st));
        st.bbox[0] = INT_MAX; st.bbox[1] = INT_MAX; st.bbox[2] = -1; st.bbox[3] = -1;
    }
    IcpBuffers B = c->B;
    B.scene = c->d_scene; B.models = c->d_models; B.model_bbox = c->d_model_bbox; B.in = c->d_in; B.st = c->d_st;
    B.count = count;
    memcpy(B.sK, c->sK, sizeof(B.sK));
    HIP_TRY(hipMemcpyAsync(c->d_in, c->h_in, (size_t)count * sizeof(IcpIn), hipMemcpyHostToDevice, c->s));
    HIP_TRY(hipMemcpyAsync(c->d_st, c->h_st, (size_t)count * sizeof(IcpState), hipMemcpyHostToDevice, c->s));
    HIP_TRY(hipEventRecord(c->e0, c->s));
    bool boxed = true;                                               // every slot's box was worked out when its image was uploaded: no k_icp_bbox
    for (int i = 0; i < count; ++i) boxed = boxed && c->slot_boxed[(size_t)(model_slots ? model_slots[i] : i)] != 0;
    launch_icp_pipeline(B, count, c->W, c->H, (flags & 0xFF) | (boxed ? 0x100 : 0), kVoxel, kMaxDist, kMaxIter, kRelTol, kKnn, c->solo_from, c->s);
    for (int pass = 0; pass < 3; ++pass) {
        HIP_TRY(hipEventRecord(c->e1, c->s));
        HIP_TRY(hipMemcpyAsync(c->h_st2, c->d_st, (size_t)count * sizeof(IcpState), hipMemcpyDeviceToHost, c->s));
        HIP_TRY(hipStreamSynchronize(c->s));
        HIP_TRY(hipGetLastError());
        if (!lm_icp_unfinished(c->h_st2, count)) break;
        // clouds the first team builds do not hold (more than 704 source points per workgroup): the builds with more points per thread;
        // what those leave too, or a team that timed out: the sliced launches
        if (pass == 2 || c->solo_from != 0) return lm_set_error(LM_ERR_HIP, "ICP: a hypothesis was left unfinished");
        if (pass == 0) launch_icp_team(B, count, 1, kMaxDist, kMaxIter, kRelTol, c->s);
        else launch_icp_evals(B, count, 0, kMaxIter + 1, kMaxDist, kMaxIter, kRelTol, c->s);
    }
    memcpy(c->h_st, c->h_st2, (size_t)count * sizeof(IcpState));
    c->last_count = count; c->last_flags = flags;
    if (device_ms) (void)hipEventElapsedTime(device_ms, c->e0, c->e1);
    for (int i = 0; i < count; ++i) {
        const IcpState& st = c->h_st[i];
        if (st.status == 2) return lm_set_error(LM_ERR_INVALID, "model depth image is empty");
        if (st.status == lm::kIcpStalled) return lm_set_error(LM_ERR_HIP, "hypothesis %d: the point kernels stalled (strips waited a second for each other)", i);
        if (st.status == 3)
            return lm_set_error(LM_ERR_INVALID, "hypothesis %d: point cloud too large for 64-bit voxel keys (depth spans tens of metres?)", i);
    }
    for (int i = 0; i < count; ++i) {
        const IcpState& st = c->h_st[i];
        lm_pose_result& o = results[i];
        memset(&o, 0, sizeof(o));
        if (st.status == 1) { o.residual = -1.f; continue; }        // LL.cpp:52-55
        lm_icp_compose_result(st, model_Rs + 9 * i, model_ts + 3 * i, &o);
    }
    return LM_OK;
}

extern "C" int64_t lm_icp_read_debug(lm_icp* c, int hypothesis, int kind, double* dst, int64_t capacity) {
    if (!c || hypothesis < 0 || hypothesis >= c->last_count) return lm_set_error(LM_ERR_INVALID, "no such hypothesis in the last run");
    if (hipSetDevice(c->device) != hipSuccess) return lm_set_error(LM_ERR_HIP, "hipSetDevice failed");
    const IcpState& st = c->h_st[hypothesis];
    const size_t off = (size_t)hypothesis * c->B.cap;
    const bool scene_mode = (c->last_flags & LM_ICP_SCENE_FROM_SCENE) != 0;
    std::vector<double> tmp;
    int64_t n = 0;
    switch (kind) {
        case 0: {   // source cloud
            n = (int64_t)st.n_src * 3; tmp.resize((size_t)n);
            if (n && hipMemcpy(tmp.data(), c->B.src + off * 3, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                return lm_set_error(LM_ERR_HIP, "read-back failed");
            break;
        }
        case 1: {   // target cloud, original (voxel) order
            n = (int64_t)st.n_tgt * 3; tmp.resize((size_t)n);
            if (n && hipMemcpy(tmp.data(), (scene_mode ? c->B.tgt : c->B.src) + off * 3, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                return lm_set_error(LM_ERR_HIP, "read-back failed");
            break;
        }
        case 2: {   // target normals, original order
            n = (int64_t)st.n_tgt * 3; tmp.resize((size_t)n);
            std::vector<double> sorted((size_t)n);
            std::vector<int> orig((size_t)st.n_tgt);
            if (n && (hipMemcpy(sorted.data(), c->B.normals + off * 3, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
                      hipMemcpy(orig.data(), c->B.tgt_orig + off, (size_t)st.n_tgt * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess))
                return lm_set_error(LM_ERR_HIP, "read-back failed");
            for (int p = 0; p < st.n_tgt; ++p)
                for (int k = 0; k < 3; ++k) tmp[(size_t)orig[p] * 3 + k] = sorted[(size_t)p * 3 + k];
            break;
        }
        case 3: {   // init_guess translation, final T (16), counters
            tmp = {st.init[0], st.init[1], st.init[2]};
            for (int k = 0; k < 16; ++k) tmp.push_back(st.T[k]);
            tmp.push_back((double)st.n_model); tmp.push_back((double)st.n_scene);
            tmp.push_back((double)st.gx); tmp.push_back((double)st.gy); tmp.push_back(st.cell);
            tmp.push_back((double)st.iterations);
            for (int k = 0; k < 8; ++k) tmp.push_back((double)st.clk[k]);
            for (int k = 0; k < 4; ++k) tmp.push_back((double)st.team_note[k]);
            tmp.push_back((double)st.n_src); tmp.push_back((double)st.n_tgt);
            for (int k = 0; k < 4; ++k) tmp.push_back((double)st.knn_clk[k]);
            for (int k = 0; k < 4; ++k) tmp.push_back((double)st.vox_clk[k]);
            tmp.push_back((double)st.n_model); tmp.push_back((double)st.n_scene);
            for (int k = 0; k < 16; ++k) tmp.push_back((double)st.sort_clk[k]);
            tmp.push_back((double)st.team_size); tmp.push_back((double)st.resume_it);
            n = (int64_t)tmp.size();
            break;
        }
        case 5: {   // cumulants of the k nearest neighbours per sorted target position [n_tgt][kIcpCovStride]
            n = (int64_t)st.n_tgt * kIcpCovStride; tmp.resize((size_t)n);
            if (n && hipMemcpy(tmp.data(), c->B.cov + off * kIcpCovStride, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                return lm_set_error(LM_ERR_HIP, "read-back failed");
            break;
        }
        case 4: {   // the slices' partial sums of the last two evaluations, [2][kIcpMaxSplit][32] (slots 29..31: shader cycles of the slice)
            n = 2 * kIcpMaxSplit * 32; tmp.resize((size_t)n);
            for (int par = 0; par < 2; ++par)
                if (hipMemcpy(tmp.data() + (size_t)par * kIcpMaxSplit * 32, c->B.partial + (((size_t)par * c->last_count + hypothesis) * kIcpMaxSplit) * 32,
                              (size_t)kIcpMaxSplit * 32 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                    return lm_set_error(LM_ERR_HIP, "read-back failed");
            break;
        }
        default: return lm_set_error(LM_ERR_INVALID, "unknown debug kind %d", kind);
    }
    if (dst && capacity > 0) memcpy(dst, tmp.data(), (size_t)std::min<int64_t>(n, capacity) * sizeof(double));
    return n;
}

// ---- the reference-shaped entry points: one shared context per device -----------------------------
namespace {
std::mutex g_mu;
std::vector<lm_icp*> g_ctx;

int shared_context(int device, lm_icp** out) {
    if (device < 0) return lm_set_error(LM_ERR_INVALID, "device %d out of range", device);
    if ((size_t)device >= g_ctx.size()) g_ctx.resize((size_t)device + 1, nullptr);
    if (!g_ctx[device]) {
        int rc = lm_icp_create(device, &g_ctx[device]);
        if (rc) return rc;
    }
    *out = g_ctx[device];
    return LM_OK;
}
}  // namespace

extern "C" int lm_pose_refine_batch(int device, const uint16_t* scene_depth, int width, int height, const float* scene_K,
                                    int count, const uint16_t* const* model_depths, const float* model_Ks, const float* model_Rs,
                                    const float* model_ts, const int32_t* detect_xy, int flags, lm_pose_result* results,
                                    float* device_ms) {
    if (!scene_depth || !scene_K || count < 0 || (count && (!model_depths || !model_Ks || !model_Rs || !model_ts || !detect_xy || !results)))
        return lm_set_error(LM_ERR_INVALID, "null argument");
    if (device_ms) *device_ms = 0.f;
    if (count == 0) return LM_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    lm_icp* c = nullptr;
    int rc = shared_context(device, &c);
    if (rc) return rc;
    if ((rc = lm_icp_set_scene(c, scene_depth, width, height, scene_K))) return rc;
    if ((rc = lm_icp_set_models(c, 0, count, model_depths))) return rc;
    return lm_icp_run(c, count, nullptr, model_Ks, model_Rs, model_ts, detect_xy, flags, results, device_ms);
}

extern "C" int lm_pose_refine(int device, const uint16_t* scene_depth, const uint16_t* model_depth, int width, int height,
                              const float* scene_K, const float* model_K, const float* model_R, const float* model_t, int detect_x,
                              int detect_y, int flags, lm_pose_result* result) {
    const uint16_t* models[1] = {model_depth};
    int32_t xy[2] = {detect_x, detect_y};
    return lm_pose_refine_batch(device, scene_depth, width, height, scene_K, 1, models, model_K, model_R, model_t, xy, flags, result,
                                nullptr);
}
