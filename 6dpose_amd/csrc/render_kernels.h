// Internal interface between render.cpp (host) and render.hip (kernels).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lm {

struct MeshDev {
    const float* v;      // [nv][3] model coordinates (mm)
    const float* n;      // [nv][3] vertex normals, may be null
    const uint8_t* c;    // [nv][3] vertex colours, may be null
    const int32_t* f;    // [nf][3]
    int nv, nf;
};
struct ViewParams {      // one view: OpenCV camera, p_cam = R v + t
    double K[9], R[9], t[3];
};
struct ProjVtx {
    double z;            // eye depth (mm)
    int sx, sy;          // screen position in 1/256 pixel
    int valid, pad;
};

void launch_project(const MeshDev& M, const ViewParams* views, int count, int scale, ProjVtx* out, hipStream_t s);
void launch_raster(const MeshDev& M, const ProjVtx* pv, int count, int Ws, int Hs, double clip_near, double clip_far,
                   unsigned long long* zbuf, hipStream_t s);
void launch_resolve_depth(const unsigned long long* zbuf, int count, int W, int H, uint16_t* depth, hipStream_t s);
void launch_resolve_rgb(const MeshDev& M, const ProjVtx* pv, const ViewParams* views, const unsigned long long* zbuf, int count, int W, int H,
                        int ssaa, float ambient, uint8_t* rgb, hipStream_t s);

}  // namespace lm
