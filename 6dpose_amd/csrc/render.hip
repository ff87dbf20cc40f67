// Depth / colour rasteriser for template generation and pose refinement: replaces the
// OpenGL renderer of the reference driver (pysixd/renderer.py:306-420, called at
// linemod_and_levelup_test.py:206-215 for training views and :352 for depth_ren of a match).
//
// The reference result depends on the OpenGL implementation, so there is nothing bit-level to match;
// this rasteriser is specified so that a numpy restatement (oracle/render_oracle.py) reproduces it
// exactly:
//   * camera: OpenCV convention, p_cam = R v + t (double, fixed operation order), u = fx x/z + cx,
//     v = fy y/z + cy; pixel (i, j) is sampled at (i, j) — the convention poseRefine back-projects with;
//   * vertices snapped to 1/256 pixel, edge functions in int64 (exact, watertight, top-left fill rule),
//     both windings drawn (the reference does not cull);
//   * depth: perspective-correct, z = 1 / sum(lambda_i / z_i) in double, nearest fragment wins, ties go
//     to the lower triangle index; stored as uint16 by truncation (depth.astype(np.uint16));
//   * colour (mode 'rgb', phong): per-fragment normal / colour interpolated perspective-correctly, light at
//     the eye, light = min(1, ambient + max(0, L.N)), rendered at ssaa x the resolution and box-averaged
//     (cv2.resize INTER_AREA with an integer factor).
// One thread per triangle and view for coverage (model triangles are smaller than a pixel at template
// distances), 64-bit atomicMin on (depth bits | triangle) for the depth test, one thread per pixel to resolve.
#include "render_kernels.h"

namespace lm {

static __device__ __forceinline__ bool top_left(long long ex, long long ey) {
    // edge vector (ex, ey) in a y-down raster: "top" = horizontal edge going left (ey == 0 && ex < 0)... with the
    // winding normalised to positive area below, an edge is top if ey == 0 && ex > 0, left if ey < 0
    return (ey == 0 && ex > 0) || ey < 0;
}

__global__ void k_project(MeshDev M, const ViewParams* __restrict__ views, int scale, ProjVtx* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, view = blockIdx.y;
    if (i >= M.nv) return;
    const ViewParams V = views[view];
    const double x = M.v[3 * (size_t)i], y = M.v[3 * (size_t)i + 1], z = M.v[3 * (size_t)i + 2];
    const double px = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(V.R[0], x), __dmul_rn(V.R[1], y)), __dmul_rn(V.R[2], z)), V.t[0]);
    const double py = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(V.R[3], x), __dmul_rn(V.R[4], y)), __dmul_rn(V.R[5], z)), V.t[1]);
    const double pz = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(V.R[6], x), __dmul_rn(V.R[7], y)), __dmul_rn(V.R[8], z)), V.t[2]);
    ProjVtx o;
    o.z = pz;
    o.valid = 0; o.sx = 0; o.sy = 0;
    if (pz > 0.0) {
        const double s = (double)scale;
        const double u = __dadd_rn(__ddiv_rn(__dmul_rn(__dmul_rn(V.K[0], s), px), pz), __dmul_rn(V.K[2], s));
        const double v = __dadd_rn(__ddiv_rn(__dmul_rn(__dmul_rn(V.K[4], s), py), pz), __dmul_rn(V.K[5], s));
        const double fu = rint(__dmul_rn(u, 256.0)), fv = rint(__dmul_rn(v, 256.0));
        if (fabs(fu) < 1e9 && fabs(fv) < 1e9) { o.sx = (int)fu; o.sy = (int)fv; o.valid = 1; }
    }
    out[(size_t)view * M.nv + i] = o;
}

// barycentric weights (unnormalised, int64) of pixel centre (px, py) [1/256 px] for the triangle with positive area
struct Tri { long long x0, y0, x1, y1, x2, y2, area; };

static __device__ __forceinline__ bool make_tri(const ProjVtx& a, const ProjVtx& b, const ProjVtx& c, Tri& t, bool& flipped) {
    long long area = ((long long)b.sx - a.sx) * ((long long)c.sy - a.sy) - ((long long)b.sy - a.sy) * ((long long)c.sx - a.sx);
    if (area == 0) return false;
    flipped = area < 0;
    t.x0 = a.sx; t.y0 = a.sy;
    if (!flipped) { t.x1 = b.sx; t.y1 = b.sy; t.x2 = c.sx; t.y2 = c.sy; t.area = area; }
    else { t.x1 = c.sx; t.y1 = c.sy; t.x2 = b.sx; t.y2 = b.sy; t.area = -area; }
    return true;
}
// w0 belongs to vertex 0 (edge 1->2), w1 to vertex 1 (edge 2->0), w2 to vertex 2 (edge 0->1); inside iff all pass the fill rule
static __device__ __forceinline__ bool weights(const Tri& t, long long px, long long py, long long& w0, long long& w1, long long& w2) {
    w0 = (t.x2 - t.x1) * (py - t.y1) - (t.y2 - t.y1) * (px - t.x1);
    w1 = (t.x0 - t.x2) * (py - t.y2) - (t.y0 - t.y2) * (px - t.x2);
    w2 = (t.x1 - t.x0) * (py - t.y0) - (t.y1 - t.y0) * (px - t.x0);
    const bool i0 = w0 > 0 || (w0 == 0 && top_left(t.x2 - t.x1, t.y2 - t.y1));
    const bool i1 = w1 > 0 || (w1 == 0 && top_left(t.x0 - t.x2, t.y0 - t.y2));
    const bool i2 = w2 > 0 || (w2 == 0 && top_left(t.x1 - t.x0, t.y1 - t.y0));
    return i0 && i1 && i2;
}

static __device__ __forceinline__ double frag_depth(const Tri& t, long long w0, long long w1, long long w2, double z0, double z1, double z2) {
    const double A = (double)t.area;
    const double l0 = __ddiv_rn((double)w0, A), l1 = __ddiv_rn((double)w1, A), l2 = __ddiv_rn((double)w2, A);
    const double s = __dadd_rn(__dadd_rn(__ddiv_rn(l0, z0), __ddiv_rn(l1, z1)), __ddiv_rn(l2, z2));
    return __ddiv_rn(1.0, s);
}

__global__ void k_raster(MeshDev M, const ProjVtx* __restrict__ pv, int Ws, int Hs, double clip_near, double clip_far,
                         unsigned long long* __restrict__ zbuf) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x, view = blockIdx.y;
    if (f >= M.nf) return;
    const ProjVtx* P = pv + (size_t)view * M.nv;
    const ProjVtx a = P[M.f[3 * (size_t)f]], b = P[M.f[3 * (size_t)f + 1]], c = P[M.f[3 * (size_t)f + 2]];
    if (!a.valid || !b.valid || !c.valid) return;                      // behind the camera: not drawn (no near-plane clipping)
    Tri t;
    bool flipped;
    if (!make_tri(a, b, c, t, flipped)) return;
    const double z0 = a.z, z1 = flipped ? c.z : b.z, z2 = flipped ? b.z : c.z;
    long long minx = min(t.x0, min(t.x1, t.x2)), maxx = max(t.x0, max(t.x1, t.x2));
    long long miny = min(t.y0, min(t.y1, t.y2)), maxy = max(t.y0, max(t.y1, t.y2));
    int ix0 = (int)max((minx + 255) >> 8, 0ll), ix1 = (int)min(maxx >> 8, (long long)Ws - 1);
    int iy0 = (int)max((miny + 255) >> 8, 0ll), iy1 = (int)min(maxy >> 8, (long long)Hs - 1);
    unsigned long long* Z = zbuf + (size_t)view * Ws * Hs;
    for (int y = iy0; y <= iy1; ++y)
        for (int x = ix0; x <= ix1; ++x) {
            long long w0, w1, w2;
            if (!weights(t, (long long)x << 8, (long long)y << 8, w0, w1, w2)) continue;
            const double z = frag_depth(t, w0, w1, w2, z0, z1, z2);
            if (!(z >= clip_near && z <= clip_far)) continue;
            const float zf = (float)z;
            const unsigned long long key = ((unsigned long long)__float_as_uint(zf) << 32) | (unsigned int)f;
            atomicMin(&Z[(size_t)y * Ws + x], key);
        }
}

__global__ void k_resolve_depth(const unsigned long long* __restrict__ zbuf, int n_per_view, uint16_t* __restrict__ depth) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_per_view * gridDim.y;
    const size_t g = (size_t)blockIdx.y * n_per_view + i;
    if (i >= (size_t)n_per_view || g >= total) return;
    const unsigned long long k = zbuf[g];
    uint16_t d = 0;
    if (k != ~0ull) {
        const float z = __uint_as_float((unsigned int)(k >> 32));
        d = z >= 65535.f ? (uint16_t)65535 : (uint16_t)z;              // astype(np.uint16): truncation
    }
    depth[g] = d;
}

// Shades the ssaa x ssaa samples of one output pixel from the id buffer and box-averages them.
__global__ void k_resolve_rgb(MeshDev M, const ProjVtx* __restrict__ pv, const ViewParams* __restrict__ views,
                              const unsigned long long* __restrict__ zbuf, int W, int H, int ssaa, float ambient, uint8_t* __restrict__ rgb) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, view = blockIdx.z;
    if (x >= W) return;
    const int Ws = W * ssaa, Hs = H * ssaa;
    const ProjVtx* P = pv + (size_t)view * M.nv;
    const ViewParams V = views[view];
    const unsigned long long* Z = zbuf + (size_t)view * Ws * Hs;
    int acc[3] = {0, 0, 0};
    for (int sy = 0; sy < ssaa; ++sy)
        for (int sx = 0; sx < ssaa; ++sx) {
            const int X = x * ssaa + sx, Y = y * ssaa + sy;
            const unsigned long long k = Z[(size_t)Y * Ws + X];
            if (k == ~0ull) continue;                                   // background: black
            const int f = (int)(unsigned int)k;
            const int i0 = M.f[3 * (size_t)f], i1 = M.f[3 * (size_t)f + 1], i2 = M.f[3 * (size_t)f + 2];
            const ProjVtx a = P[i0], b = P[i1], c = P[i2];
            Tri t;
            bool flipped;
            if (!make_tri(a, b, c, t, flipped)) continue;
            long long w0, w1, w2;
            (void)weights(t, (long long)X << 8, (long long)Y << 8, w0, w1, w2);
            const int j1 = flipped ? i2 : i1, j2 = flipped ? i1 : i2;
            const double z0 = a.z, z1 = flipped ? c.z : b.z, z2 = flipped ? b.z : c.z;
            const double z = frag_depth(t, w0, w1, w2, z0, z1, z2);
            // perspective-correct attribute weights
            const double A = (double)t.area;
            const float q0 = (float)((double)w0 / A / z0 * z), q1 = (float)((double)w1 / A / z1 * z), q2 = (float)((double)w2 / A / z2 * z);
            float nx = 0.f, ny = 0.f, nz = -1.f;
            if (M.n) {
                const float n0[3] = {M.n[3 * (size_t)i0], M.n[3 * (size_t)i0 + 1], M.n[3 * (size_t)i0 + 2]};
                const float n1[3] = {M.n[3 * (size_t)j1], M.n[3 * (size_t)j1 + 1], M.n[3 * (size_t)j1 + 2]};
                const float n2[3] = {M.n[3 * (size_t)j2], M.n[3 * (size_t)j2 + 1], M.n[3 * (size_t)j2 + 2]};
                const float mx = q0 * n0[0] + q1 * n1[0] + q2 * n2[0], my = q0 * n0[1] + q1 * n1[1] + q2 * n2[1],
                            mz = q0 * n0[2] + q1 * n1[2] + q2 * n2[2];
                // to eye space with the rotation (normal matrix of a rigid transform)
                nx = (float)V.R[0] * mx + (float)V.R[1] * my + (float)V.R[2] * mz;
                ny = (float)V.R[3] * mx + (float)V.R[4] * my + (float)V.R[5] * mz;
                nz = (float)V.R[6] * mx + (float)V.R[7] * my + (float)V.R[8] * mz;
            }
            const float nl = sqrtf(nx * nx + ny * ny + nz * nz);
            // fragment position in eye space from the pixel ray
            const float fz = (float)z;
            const float ex = ((float)X - (float)(V.K[2] * ssaa)) / (float)(V.K[0] * ssaa) * fz;
            const float ey = ((float)Y - (float)(V.K[5] * ssaa)) / (float)(V.K[4] * ssaa) * fz;
            const float el = sqrtf(ex * ex + ey * ey + fz * fz);
            float diff = 0.f;
            if (nl > 0.f && el > 0.f) diff = -(ex * nx + ey * ny + fz * nz) / (el * nl);   // L = -p/|p| (light at the eye)
            if (diff < 0.f) diff = 0.f;
            float lw = ambient + diff;
            if (lw > 1.f) lw = 1.f;
            float col[3] = {0.5f, 0.5f, 0.5f};                          // renderer.py:331 default colour
            if (M.c) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    col[ch] = (q0 * M.c[3 * (size_t)i0 + ch] + q1 * M.c[3 * (size_t)j1 + ch] + q2 * M.c[3 * (size_t)j2 + ch]) * (1.f / 255.f);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float v = lw * col[ch] * 255.f;
                v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                acc[ch] += (int)rintf(v);
            }
        }
    const int n = ssaa * ssaa;
    uint8_t* o = rgb + (((size_t)view * H + y) * W + x) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[ch] = (uint8_t)((2 * acc[ch] + n) / (2 * n));   // rounded mean (INTER_AREA)
}

void launch_project(const MeshDev& M, const ViewParams* views, int count, int scale, ProjVtx* out, hipStream_t s) {
    if (count <= 0 || M.nv <= 0) return;
    hipLaunchKernelGGL(k_project, dim3((M.nv + 255) / 256, count), dim3(256), 0, s, M, views, scale, out);
}
void launch_raster(const MeshDev& M, const ProjVtx* pv, int count, int Ws, int Hs, double clip_near, double clip_far,
                   unsigned long long* zbuf, hipStream_t s) {
    if (count <= 0 || M.nf <= 0) return;
    (void)hipMemsetAsync(zbuf, 0xFF, (size_t)count * Ws * Hs * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_raster, dim3((M.nf + 255) / 256, count), dim3(256), 0, s, M, pv, Ws, Hs, clip_near, clip_far, zbuf);
}
void launch_resolve_depth(const unsigned long long* zbuf, int count, int W, int H, uint16_t* depth, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_resolve_depth, dim3((W * H + 255) / 256, count), dim3(256), 0, s, zbuf, W * H, depth);
}
void launch_resolve_rgb(const MeshDev& M, const ProjVtx* pv, const ViewParams* views, const unsigned long long* zbuf, int count, int W, int H,
                        int ssaa, float ambient, uint8_t* rgb, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_resolve_rgb, dim3((W + 255) / 256, H, count), dim3(256), 0, s, M, pv, views, zbuf, W, H, ssaa, ambient, rgb);
}

}  // namespace lm
