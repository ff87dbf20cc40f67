// Private view of lm_mesh shared by render.cpp, detector.cpp and pipeline.cpp (not part of the C ABI).
#pragma once
#include "../../include/amd_linemod.h"
#include "render_kernels.h"

int lm_set_error(int code, const char* fmt, ...);
#ifndef HIP_TRY
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return lm_set_error(LM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                                  __FILE__, __LINE__);                                        \
    } while (0)
#endif

struct lm_mesh {
    int device = 0;
    int nv = 0, nf = 0;
    hipStream_t s = nullptr;
    float* d_v = nullptr;
    float* d_n = nullptr;
    uint8_t* d_c = nullptr;
    int32_t* d_f = nullptr;
    // render scratch / results of the last call (device)
    lm::ViewParams* d_views = nullptr;
    lm::ProjVtx* d_pv = nullptr;
    unsigned long long* d_zbuf = nullptr;
    uint16_t* d_depth = nullptr;
    uint8_t* d_rgb = nullptr;
    size_t cap_views = 0, cap_pv = 0, cap_zbuf = 0, cap_depth = 0, cap_rgb = 0;
    int last_W = 0, last_H = 0, last_count = 0;
};

// Renders `count` views into the mesh's device buffers (d_depth [count][H][W], d_rgb [count][H][W][3]) on m->s; no host copy.
int lm_mesh_render_device(lm_mesh* m, int count, int W, int H, const float* Ks, const float* Rs, const float* ts, float clip_near,
                          float clip_far, float ambient, int ssaa, bool want_depth, bool want_rgb);
