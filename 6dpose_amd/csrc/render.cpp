// lm_mesh — triangle meshes resident in HBM and their rendering into depth / colour images:
// the job pysixd's OpenGL renderer does for the reference driver (linemod_and_levelup_test.py:206-215 training
// views, :352 depth_ren of a match).  PLY reading follows pysixd/inout.py:load_ply (ascii and
// binary_little_endian; x y z [nx ny nz] [red green blue], triangular faces).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "render_internal.h"

using namespace lm;

extern "C" int lm_mesh_create(int device, const float* vertices, const float* normals, const uint8_t* colors, int nv, const int32_t* faces,
                              int nf, lm_mesh** out) {
    if (!out || !vertices || !faces || nv <= 0 || nf <= 0) return lm_set_error(LM_ERR_INVALID, "null / empty mesh");
    *out = nullptr;
    for (int i = 0; i < 3 * nf; ++i)
        if (faces[i] < 0 || faces[i] >= nv) return lm_set_error(LM_ERR_INVALID, "face index %d out of range", faces[i]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lm_set_error(LM_ERR_NO_DEVICE, "no HIP device visible; libamdlinemod has no CPU fallback");
    if (device < 0 || device >= ndev) return lm_set_error(LM_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    lm_mesh* m = new lm_mesh();
    m->device = device; m->nv = nv; m->nf = nf;
    auto fail = [&](const char* what) { lm_mesh_destroy(m); return lm_set_error(LM_ERR_HIP, "%s failed", what); };
    if (hipStreamCreateWithFlags(&m->s, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate");
    if (hipMalloc((void**)&m->d_v, (size_t)nv * 3 * sizeof(float)) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&m->d_f, (size_t)nf * 3 * sizeof(int32_t)) != hipSuccess) return fail("hipMalloc");
    if (hipMemcpy(m->d_v, vertices, (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
    if (hipMemcpy(m->d_f, faces, (size_t)nf * 3 * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
    if (normals) {
        if (hipMalloc((void**)&m->d_n, (size_t)nv * 3 * sizeof(float)) != hipSuccess) return fail("hipMalloc");
        if (hipMemcpy(m->d_n, normals, (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
    }
    if (colors) {
        if (hipMalloc((void**)&m->d_c, (size_t)nv * 3) != hipSuccess) return fail("hipMalloc");
        if (hipMemcpy(m->d_c, colors, (size_t)nv * 3, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
    }
    *out = m;
    return LM_OK;
}

extern "C" void lm_mesh_destroy(lm_mesh* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->s) (void)hipStreamSynchronize(m->s);
    void* p[] = {m->d_v, m->d_n, m->d_c, m->d_f, m->d_views, m->d_pv, m->d_zbuf, m->d_depth, m->d_rgb};
    for (void* q : p) if (q) (void)hipFree(q);
    if (m->s) (void)hipStreamDestroy(m->s);
    delete m;
}

extern "C" int lm_mesh_counts(const lm_mesh* m, int* nv, int* nf) {
    if (!m) return lm_set_error(LM_ERR_INVALID, "null mesh");
    if (nv) *nv = m->nv;
    if (nf) *nf = m->nf;
    return LM_OK;
}

// ---- PLY (pysixd/inout.py:load_ply) -----------------------------------------------------------------
namespace {
struct Prop { std::string name, type, list_count, list_item; bool is_list = false; };
int type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
double read_bin(const unsigned char* p, const std::string& t) {
    if (t == "char" || t == "int8") return (double)*(const signed char*)p;
    if (t == "uchar" || t == "uint8") return (double)*p;
    if (t == "short" || t == "int16") { int16_t v; memcpy(&v, p, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, p, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; memcpy(&v, p, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; memcpy(&v, p, 4); return v; }
    if (t == "float" || t == "float32") { float v; memcpy(&v, p, 4); return v; }
    double v; memcpy(&v, p, 8); return v;
}
}  // namespace

extern "C" int lm_mesh_load_ply(int device, const char* path, lm_mesh** out) {
    if (!path || !out) return lm_set_error(LM_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return lm_set_error(LM_ERR_IO, "cannot open %s", path);
    std::vector<unsigned char> buf;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        if (sz > 0 && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return lm_set_error(LM_ERR_IO, "short read of %s", path); }
        fclose(f);
    }
    size_t pos = 0;
    auto next_line = [&](std::string& line) {
        if (pos >= buf.size()) return false;
        size_t e = pos;
        while (e < buf.size() && buf[e] != '\n') ++e;
        line.assign((const char*)&buf[pos], e - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!next_line(line) || line != "ply") return lm_set_error(LM_ERR_IO, "%s is not a PLY file", path);
    bool ascii = true;
    int nv = 0, nf = 0;
    std::vector<Prop> vprops, fprops;
    std::string cur;
    while (next_line(line)) {
        if (line == "end_header") break;
        char a[64] = {0}, b[64] = {0}, c[64] = {0}, d[64] = {0}, e[64] = {0};
        int n = sscanf(line.c_str(), "%63s %63s %63s %63s %63s", a, b, c, d, e);
        if (n >= 2 && !strcmp(a, "format")) {
            if (!strcmp(b, "ascii")) ascii = true;
            else if (!strcmp(b, "binary_little_endian")) ascii = false;
            else return lm_set_error(LM_ERR_IO, "unsupported PLY format '%s'", b);
        } else if (n >= 3 && !strcmp(a, "element")) {
            cur = b;
            if (cur == "vertex") nv = atoi(c);
            else if (cur == "face") nf = atoi(c);
        } else if (n >= 3 && !strcmp(a, "property")) {
            Prop p;
            if (!strcmp(b, "list")) { p.is_list = true; p.list_count = c; p.list_item = d; p.name = e; }
            else { p.type = b; p.name = c; }
            if (cur == "vertex") vprops.push_back(p);
            else if (cur == "face") fprops.push_back(p);
        }
    }
    if (nv <= 0 || nf <= 0) return lm_set_error(LM_ERR_IO, "%s: no vertices / faces", path);
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, ir = -1, ig = -1, ib = -1;
    for (size_t i = 0; i < vprops.size(); ++i) {
        const std::string& nm = vprops[i].name;
        if (vprops[i].is_list) return lm_set_error(LM_ERR_IO, "%s: list property on vertices", path);
        if (nm == "x") ix = (int)i; else if (nm == "y") iy = (int)i; else if (nm == "z") iz = (int)i;
        else if (nm == "nx") inx = (int)i; else if (nm == "ny") iny = (int)i; else if (nm == "nz") inz = (int)i;
        else if (nm == "red") ir = (int)i; else if (nm == "green") ig = (int)i; else if (nm == "blue") ib = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return lm_set_error(LM_ERR_IO, "%s: vertices without x/y/z", path);
    const bool has_n = inx >= 0 && iny >= 0 && inz >= 0, has_c = ir >= 0 && ig >= 0 && ib >= 0;
    std::vector<float> V((size_t)nv * 3), N(has_n ? (size_t)nv * 3 : 0);
    std::vector<uint8_t> C(has_c ? (size_t)nv * 3 : 0);
    std::vector<int32_t> F;
    F.reserve((size_t)nf * 3);
    std::vector<double> vals(vprops.size());
    if (ascii) {
        const char* p = (const char*)buf.data() + pos;
        const char* end = (const char*)buf.data() + buf.size();
        auto num = [&](double& v) {
            while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
            if (p >= end) return false;
            char* q = nullptr;
            v = strtod(p, &q);
            if (q == p) return false;
            p = q;
            return true;
        };
        for (int i = 0; i < nv; ++i) {
            for (size_t k = 0; k < vprops.size(); ++k)
                if (!num(vals[k])) return lm_set_error(LM_ERR_IO, "%s: truncated vertex list", path);
            V[3 * (size_t)i] = (float)vals[ix]; V[3 * (size_t)i + 1] = (float)vals[iy]; V[3 * (size_t)i + 2] = (float)vals[iz];
            if (has_n) { N[3 * (size_t)i] = (float)vals[inx]; N[3 * (size_t)i + 1] = (float)vals[iny]; N[3 * (size_t)i + 2] = (float)vals[inz]; }
            if (has_c) { C[3 * (size_t)i] = (uint8_t)vals[ir]; C[3 * (size_t)i + 1] = (uint8_t)vals[ig]; C[3 * (size_t)i + 2] = (uint8_t)vals[ib]; }
        }
        for (int i = 0; i < nf; ++i) {
            for (const Prop& fp : fprops) {
                double cnt;
                if (!fp.is_list) { if (!num(cnt)) return lm_set_error(LM_ERR_IO, "%s: truncated face list", path); continue; }
                if (!num(cnt)) return lm_set_error(LM_ERR_IO, "%s: truncated face list", path);
                std::vector<int32_t> idx((size_t)cnt);
                for (int k = 0; k < (int)cnt; ++k) { double v; if (!num(v)) return lm_set_error(LM_ERR_IO, "%s: truncated face list", path); idx[k] = (int32_t)v; }
                if (fp.name == "vertex_indices" || fp.name == "vertex_index") {
                    if ((int)cnt != 3) return lm_set_error(LM_ERR_IO, "%s: only triangular faces are supported (inout.py:load_ply)", path);
                    F.insert(F.end(), idx.begin(), idx.end());
                }
            }
        }
    } else {
        const unsigned char* p = buf.data() + pos;
        const unsigned char* end = buf.data() + buf.size();
        for (int i = 0; i < nv; ++i) {
            for (size_t k = 0; k < vprops.size(); ++k) {
                const int ts = type_size(vprops[k].type);
                if (!ts || p + ts > end) return lm_set_error(LM_ERR_IO, "%s: truncated / unknown vertex data", path);
                vals[k] = read_bin(p, vprops[k].type);
                p += ts;
            }
            V[3 * (size_t)i] = (float)vals[ix]; V[3 * (size_t)i + 1] = (float)vals[iy]; V[3 * (size_t)i + 2] = (float)vals[iz];
            if (has_n) { N[3 * (size_t)i] = (float)vals[inx]; N[3 * (size_t)i + 1] = (float)vals[iny]; N[3 * (size_t)i + 2] = (float)vals[inz]; }
            if (has_c) { C[3 * (size_t)i] = (uint8_t)vals[ir]; C[3 * (size_t)i + 1] = (uint8_t)vals[ig]; C[3 * (size_t)i + 2] = (uint8_t)vals[ib]; }
        }
        for (int i = 0; i < nf; ++i) {
            for (const Prop& fp : fprops) {
                if (!fp.is_list) { const int ts = type_size(fp.type); if (!ts || p + ts > end) return lm_set_error(LM_ERR_IO, "%s: truncated face data", path); p += ts; continue; }
                const int cs = type_size(fp.list_count), is = type_size(fp.list_item);
                if (!cs || !is || p + cs > end) return lm_set_error(LM_ERR_IO, "%s: truncated face data", path);
                const int cnt = (int)read_bin(p, fp.list_count);
                p += cs;
                if (cnt < 0 || p + (size_t)cnt * is > end) return lm_set_error(LM_ERR_IO, "%s: truncated face data", path);
                const bool vi = fp.name == "vertex_indices" || fp.name == "vertex_index";
                if (vi && cnt != 3) return lm_set_error(LM_ERR_IO, "%s: only triangular faces are supported (inout.py:load_ply)", path);
                for (int k = 0; k < cnt; ++k) { if (vi) F.push_back((int32_t)read_bin(p, fp.list_item)); p += is; }
            }
        }
    }
    if ((int)(F.size() / 3) != nf) return lm_set_error(LM_ERR_IO, "%s: %d faces announced, %d read", path, nf, (int)(F.size() / 3));
    return lm_mesh_create(device, V.data(), has_n ? N.data() : nullptr, has_c ? C.data() : nullptr, nv, F.data(), nf, out);
}

// ---- rendering -----------------------------------------------------------------------------------------
int lm_mesh_render_device(lm_mesh* m, int count, int W, int H, const float* Ks, const float* Rs, const float* ts, float clip_near,
                          float clip_far, float ambient, int ssaa, bool want_depth, bool want_rgb) {
    if (!m || count <= 0 || W <= 0 || H <= 0 || !Ks || !Rs || !ts) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (ssaa < 1 || ssaa > 8) return lm_set_error(LM_ERR_INVALID, "ssaa must be in 1..8");
    HIP_TRY(hipSetDevice(m->device));
    const size_t nview = (size_t)count;
    std::vector<ViewParams> hv(nview);
    for (int i = 0; i < count; ++i) {
        for (int k = 0; k < 9; ++k) { hv[i].K[k] = Ks[9 * i + k]; hv[i].R[k] = Rs[9 * i + k]; }
        for (int k = 0; k < 3; ++k) hv[i].t[k] = ts[3 * i + k];
    }
    auto ensure = [&](void** p, size_t& cap, size_t bytes) -> int {
        if (bytes <= cap) return LM_OK;
        if (*p) (void)hipFree(*p);
        *p = nullptr; cap = 0;
        HIP_TRY(hipMalloc(p, bytes));
        cap = bytes;
        return LM_OK;
    };
    const int smax = want_rgb ? ssaa : 1;
    int rc;
    if ((rc = ensure((void**)&m->d_views, m->cap_views, nview * sizeof(ViewParams)))) return rc;
    if ((rc = ensure((void**)&m->d_pv, m->cap_pv, nview * m->nv * sizeof(ProjVtx)))) return rc;
    if ((rc = ensure((void**)&m->d_zbuf, m->cap_zbuf, nview * W * H * smax * smax * sizeof(unsigned long long)))) return rc;
    if (want_depth && (rc = ensure((void**)&m->d_depth, m->cap_depth, nview * W * H * sizeof(uint16_t)))) return rc;
    if (want_rgb && (rc = ensure((void**)&m->d_rgb, m->cap_rgb, nview * W * H * 3))) return rc;
    HIP_TRY(hipMemcpyAsync(m->d_views, hv.data(), nview * sizeof(ViewParams), hipMemcpyHostToDevice, m->s));
    HIP_TRY(hipStreamSynchronize(m->s));                              // hv is a local
    MeshDev M{m->d_v, m->d_n, m->d_c, m->d_f, m->nv, m->nf};
    if (want_depth) {                                                 // the driver renders depth at the native resolution (:206-209)
        launch_project(M, m->d_views, count, 1, m->d_pv, m->s);
        launch_raster(M, m->d_pv, count, W, H, clip_near, clip_far, m->d_zbuf, m->s);
        launch_resolve_depth(m->d_zbuf, count, W, H, m->d_depth, m->s);
    }
    if (want_rgb) {                                                   // and colour at ssaa x, box-filtered (:212-216)
        launch_project(M, m->d_views, count, ssaa, m->d_pv, m->s);
        launch_raster(M, m->d_pv, count, W * ssaa, H * ssaa, clip_near, clip_far, m->d_zbuf, m->s);
        launch_resolve_rgb(M, m->d_pv, m->d_views, m->d_zbuf, count, W, H, ssaa, ambient, m->d_rgb, m->s);
    }
    HIP_TRY(hipGetLastError());
    m->last_W = W; m->last_H = H; m->last_count = count;
    return LM_OK;
}

extern "C" int lm_mesh_render(lm_mesh* m, int count, int width, int height, const float* Ks, const float* Rs, const float* ts,
                              float clip_near, float clip_far, float ambient, int ssaa, uint16_t* depth_out, uint8_t* rgb_out) {
    if (!depth_out && !rgb_out) return lm_set_error(LM_ERR_INVALID, "nothing to render into");
    // views in chunks that keep the ssaa id buffer below ~2 GB
    const size_t per_view = (size_t)width * height * (rgb_out ? (size_t)ssaa * ssaa : 1) * sizeof(unsigned long long);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)2 << 30) / std::max<size_t>(per_view, 1)));
    for (int c0 = 0; c0 < count; c0 += chunk) {
        const int n = std::min(chunk, count - c0);
        int rc = lm_mesh_render_device(m, n, width, height, Ks + 9 * (size_t)c0, Rs + 9 * (size_t)c0, ts + 3 * (size_t)c0, clip_near, clip_far,
                                       ambient, ssaa, depth_out != nullptr, rgb_out != nullptr);
        if (rc) return rc;
        const size_t npx = (size_t)width * height;
        if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out + (size_t)c0 * npx, m->d_depth, (size_t)n * npx * sizeof(uint16_t), hipMemcpyDeviceToHost, m->s));
        if (rgb_out) HIP_TRY(hipMemcpyAsync(rgb_out + (size_t)c0 * npx * 3, m->d_rgb, (size_t)n * npx * 3, hipMemcpyDeviceToHost, m->s));
        HIP_TRY(hipStreamSynchronize(m->s));
    }
    return LM_OK;
}
