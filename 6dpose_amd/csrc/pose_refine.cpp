// poseRefine::process (LL.cpp:27-155) behind the C ABI: host-side cloud preparation (dilate, bbox,
// back-projection, centroid init, voxel down-sampling — a few thousand points, sequential
// bookkeeping) and the GPU ICP (icp.hip: kNN normals + the whole point-to-plane loop in one launch,
// one workgroup per hypothesis).  Deterministic rules shared with oracle/linemod_oracle.py are
// listed in DESIGN.md §ICP.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/amd_linemod.h"
#include "icp_kernels.h"

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
constexpr int kDilate = 4;            // LL.cpp:45

struct P3 { double x, y, z; };

struct Prepared {
    bool rejected = false;            // residual = -1 (LL.cpp:52-55)
    std::vector<P3> src, tgt;         // voxel-down-sampled clouds
    double init[16];
    float base[16];                   // init_base (float, LL.cpp:34-41)
};

// open3d PointCloud::VoxelDownSample: mean per voxel; output in ascending (ix,iy,iz) order, points of
// one voxel summed in input order.
std::vector<P3> voxel_down_sample(const std::vector<P3>& pts, double voxel) {
    std::vector<P3> out;
    if (pts.empty()) return out;
    double mnx = pts[0].x, mny = pts[0].y, mnz = pts[0].z;
    for (const P3& p : pts) { mnx = std::min(mnx, p.x); mny = std::min(mny, p.y); mnz = std::min(mnz, p.z); }
    mnx -= voxel * 0.5; mny -= voxel * 0.5; mnz -= voxel * 0.5;
    struct Key { long long ix, iy, iz; int idx; };
    std::vector<Key> keys(pts.size());
    for (size_t i = 0; i < pts.size(); ++i) {
        keys[i].ix = (long long)floor((pts[i].x - mnx) / voxel);
        keys[i].iy = (long long)floor((pts[i].y - mny) / voxel);
        keys[i].iz = (long long)floor((pts[i].z - mnz) / voxel);
        keys[i].idx = (int)i;
    }
    std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        if (a.ix != b.ix) return a.ix < b.ix;
        if (a.iy != b.iy) return a.iy < b.iy;
        return a.iz < b.iz;
    });
    size_t a = 0;
    while (a < keys.size()) {
        size_t b = a;
        double sx = 0, sy = 0, sz = 0;
        while (b < keys.size() && keys[b].ix == keys[a].ix && keys[b].iy == keys[a].iy && keys[b].iz == keys[a].iz) {
            const P3& p = pts[keys[b].idx];
            sx += p.x; sy += p.y; sz += p.z;
            ++b;
        }
        double n = (double)(b - a);
        out.push_back(P3{sx / n, sy / n, sz / n});
        a = b;
    }
    return out;
}

// LL.cpp:34-109 for one hypothesis
int prepare(const uint16_t* scene, const uint16_t* model, int W, int H, const float* sK, const float* mK, const float* R,
            const float* t, int dx, int dy, int flags, Prepared& out) {
    // init_base (float): [R|t], only t.z / 1000 (LL.cpp:34-39)
    float* B = out.base;
    for (int i = 0; i < 16; ++i) B[i] = 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) B[4 * r + c] = R[3 * r + c];
        B[4 * r + 3] = t[r];
    }
    B[11] = B[11] / 1000.0f;
    B[15] = 1.f;
    // modelMask = dilate(modelDepth > 0, 9x9) ; bbox (LL.cpp:43-50)
    std::vector<uint8_t> m0((size_t)W * H), mrow((size_t)W * H), mask((size_t)W * H);
    for (size_t i = 0; i < m0.size(); ++i) m0[i] = model[i] > 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v = 0;
            for (int k = std::max(0, x - kDilate); k <= std::min(W - 1, x + kDilate) && !v; ++k) v |= m0[(size_t)y * W + k];
            mrow[(size_t)y * W + x] = v;
        }
    int bx0 = W, by0 = H, bx1 = -1, by1 = -1;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v = 0;
            for (int k = std::max(0, y - kDilate); k <= std::min(H - 1, y + kDilate) && !v; ++k) v |= mrow[(size_t)k * W + x];
            mask[(size_t)y * W + x] = v;
            if (v) { bx0 = std::min(bx0, x); bx1 = std::max(bx1, x); by0 = std::min(by0, y); by1 = std::max(by1, y); }
        }
    if (bx1 < 0) return lm_set_error(LM_ERR_INVALID, "model depth image is empty");
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    if (dx + bw >= W || dy + bh >= H) { out.rejected = true; return LM_OK; }   // LL.cpp:52-55
    const double anchor = model[(size_t)(H / 2) * W + W / 2] / 1000.0;          // LL.cpp:62
    std::vector<P3> mp, sp;
    double cmx = 0, cmy = 0, cmz = 0, csx = 0, csy = 0, csz = 0;
    long cs_n = 0;
    for (int r = 0; r < bh; ++r)
        for (int c = 0; c < bw; ++c) {
            int mr = r + by0, mc = c + bx0;
            int sr = std::max(r + dy - kDilate, 0), sc = std::max(c + dx - kDilate, 0);
            if (!mask[(size_t)mr * W + mc]) continue;
            uint16_t md = model[(size_t)mr * W + mc];
            if (md > 0) {
                double z = md / 1000.0;
                // (int - float) / float evaluated in float, then * double (LL.cpp:79-80)
                double x = (double)(((float)mc - mK[2]) / mK[0]) * z;
                double y = (double)(((float)mr - mK[5]) / mK[4]) * z;
                mp.push_back(P3{x, y, z});
                cmx += x; cmy += y; cmz += z;
            }
            uint16_t sd = scene[(size_t)sr * W + sc];
            if (sd > 0) {
                double z = sd / 1000.0;
                double x = (double)(((float)sc - sK[2]) / sK[0]) * z;
                double y = (double)(((float)sr - sK[5]) / sK[4]) * z;
                sp.push_back(P3{x, y, z});
                if (fabs(z - anchor) < 0.4 && md > 0) { csx += x; csy += y; csz += z; ++cs_n; }
            }
        }
    const double nm = (double)mp.size();
    for (int i = 0; i < 16; ++i) out.init[i] = (i % 5 == 0) ? 1.0 : 0.0;
    out.init[3] = csx / (double)cs_n - cmx / nm;     // NaN when cs_n == 0, as the reference (LL.cpp:101-104)
    out.init[7] = csy / (double)cs_n - cmy / nm;
    out.init[11] = csz / (double)cs_n - cmz / nm;
    out.src = voxel_down_sample(mp, kVoxel);                                              // LL.cpp:108
    out.tgt = voxel_down_sample((flags & LM_ICP_SCENE_FROM_SCENE) ? sp : mp, kVoxel);    // LL.cpp:109 (sic: model)
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
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lm_set_error(LM_ERR_NO_DEVICE, "no HIP device visible; libamdlinemod has no CPU fallback");
    if (device < 0 || device >= ndev) return lm_set_error(LM_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));

    std::vector<Prepared> prep((size_t)count);
    std::vector<IcpProblem> probs;
    std::vector<int> prob_of((size_t)count, -1);
    std::vector<double> arena;
    int max_tgt = 0;
    for (int i = 0; i < count; ++i) {
        if (!model_depths[i]) return lm_set_error(LM_ERR_INVALID, "model depth %d is null", i);
        int rc = prepare(scene_depth, model_depths[i], width, height, scene_K, model_Ks + 9 * i, model_Rs + 9 * i, model_ts + 3 * i,
                         detect_xy[2 * i], detect_xy[2 * i + 1], flags, prep[i]);
        if (rc) return rc;
        if (prep[i].rejected) continue;
        IcpProblem pb{};
        pb.src_off = (int)(arena.size() / 3); pb.n_src = (int)prep[i].src.size();
        for (const P3& p : prep[i].src) { arena.push_back(p.x); arena.push_back(p.y); arena.push_back(p.z); }
        pb.tgt_off = (int)(arena.size() / 3); pb.n_tgt = (int)prep[i].tgt.size();
        for (const P3& p : prep[i].tgt) { arena.push_back(p.x); arena.push_back(p.y); arena.push_back(p.z); }
        memcpy(pb.init, prep[i].init, sizeof(pb.init));
        max_tgt = std::max(max_tgt, pb.n_tgt);
        prob_of[i] = (int)probs.size();
        probs.push_back(pb);
    }
    std::vector<IcpResult> res(probs.size());
    float ms = 0.f;
    if (!probs.empty()) {
        double *d_pts = nullptr, *d_nrm = nullptr, *d_work = nullptr;
        IcpProblem* d_probs = nullptr;
        IcpResult* d_res = nullptr;
        hipStream_t s = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const size_t ab = std::max<size_t>(arena.size(), 3) * sizeof(double);
        int rc = LM_OK;
        auto cleanup = [&]() {
            if (d_pts) (void)hipFree(d_pts); if (d_nrm) (void)hipFree(d_nrm); if (d_work) (void)hipFree(d_work);
            if (d_probs) (void)hipFree(d_probs); if (d_res) (void)hipFree(d_res);
            if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
            if (s) (void)hipStreamDestroy(s);
        };
#define TRY_OR_CLEAN(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return lm_set_error(LM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
        TRY_OR_CLEAN(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        TRY_OR_CLEAN(hipEventCreate(&e0));
        TRY_OR_CLEAN(hipEventCreate(&e1));
        TRY_OR_CLEAN(hipMalloc((void**)&d_pts, ab));
        TRY_OR_CLEAN(hipMalloc((void**)&d_nrm, ab));
        TRY_OR_CLEAN(hipMalloc((void**)&d_work, ab));
        TRY_OR_CLEAN(hipMalloc((void**)&d_probs, probs.size() * sizeof(IcpProblem)));
        TRY_OR_CLEAN(hipMalloc((void**)&d_res, probs.size() * sizeof(IcpResult)));
        TRY_OR_CLEAN(hipMemcpyAsync(d_pts, arena.data(), arena.size() * sizeof(double), hipMemcpyHostToDevice, s));
        TRY_OR_CLEAN(hipMemsetAsync(d_nrm, 0, ab, s));
        TRY_OR_CLEAN(hipMemcpyAsync(d_probs, probs.data(), probs.size() * sizeof(IcpProblem), hipMemcpyHostToDevice, s));
        TRY_OR_CLEAN(hipEventRecord(e0, s));
        launch_knn_normals(d_pts, d_nrm, d_probs, (int)probs.size(), max_tgt, kKnn, s);        // LL.cpp:127
        launch_icp(d_pts, d_nrm, d_work, d_probs, d_res, (int)probs.size(), kMaxDist, kMaxIter, kRelTol, s);   // LL.cpp:128-130
        TRY_OR_CLEAN(hipEventRecord(e1, s));
        TRY_OR_CLEAN(hipMemcpyAsync(res.data(), d_res, res.size() * sizeof(IcpResult), hipMemcpyDeviceToHost, s));
        TRY_OR_CLEAN(hipStreamSynchronize(s));
        TRY_OR_CLEAN(hipGetLastError());
        (void)hipEventElapsedTime(&ms, e0, e1);
        cleanup();
        (void)rc;
#undef TRY_OR_CLEAN
    }
    if (device_ms) *device_ms = ms;
    for (int i = 0; i < count; ++i) {
        lm_pose_result& o = results[i];
        memset(&o, 0, sizeof(o));
        if (prep[i].rejected) { o.residual = -1.f; continue; }
        const IcpResult& r = res[prob_of[i]];
        // result = transformation_ * init_base.cast<double>() (LL.cpp:146); t * 1000 (LL.cpp:154)
        double M[16];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                double v = 0;
                for (int k = 0; k < 4; ++k) v += r.T[4 * a + k] * (double)prep[i].base[4 * k + b];
                M[4 * a + b] = v;
            }
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) o.R[3 * a + b] = M[4 * a + b];
            o.t[a] = M[4 * a + 3] * 1000.0;
        }
        o.residual = (float)r.fitness;       // residual = fitness_ (LL.cpp:148)
        o.inlier_rmse = (float)r.rmse;
        o.iterations = r.iterations;
        o.n_source = (int)prep[i].src.size();
        o.n_target = (int)prep[i].tgt.size();
    }
    return LM_OK;
}

extern "C" int lm_pose_refine(int device, const uint16_t* scene_depth, const uint16_t* model_depth, int width, int height,
                              const float* scene_K, const float* model_K, const float* model_R, const float* model_t, int detect_x,
                              int detect_y, int flags, lm_pose_result* result) {
    const uint16_t* models[1] = {model_depth};
    int32_t xy[2] = {detect_x, detect_y};
    return lm_pose_refine_batch(device, scene_depth, width, height, scene_K, 1, models, model_K, model_R, model_t, xy, flags, result,
                                nullptr);
}
