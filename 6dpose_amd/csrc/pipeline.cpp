// lm_pipeline — the per-frame loop of the reference driver (linemod_and_levelup_test.py:324-372) as ONE
// stream of device work: Detector::match (front end, coarse, refinement) -> boxes of the matched
// templates -> greedy NMS -> the first top_k kept detections -> poseRefine on each, with the rendered
// depth of every template view resident in HBM.  The host enqueues everything, waits once, and composes
// [R|t] = T_icp * [R_view | t_view] (LL.cpp:146-154).  SURVEY §8f N1.
//
// Equals, on the same frame: lm_detector_match -> lm_nms_boxes on (x, y, x+width, y+height, similarity)
// -> lm_pose_refine_batch on the first top_k kept matches (tests/test_gpu_parity.py checks exactly that).
#include <string.h>

#include <algorithm>
#include <map>

#include "detector_internal.h"
#include "icp_internal.h"
#include "render_internal.h"

namespace {
constexpr double kVoxel = 0.0025, kMaxDist = 0.01, kRelTol = 1e-6;   // as pose_refine.cpp (LL.cpp:106, :31; Open3D defaults)
constexpr int kMaxIter = 30, kKnn = 30;
}

struct lm_pipeline {
    lm_detector* det = nullptr;
    lm_icp* icp = nullptr;
    int W = 0, H = 0;
    struct ClassViews { int base = 0, count = 0; };
    std::map<std::string, ClassViews> views;      // class id -> range of view slots
    int num_views = 0;                            // slots handed out
    std::vector<float> view_K, view_R, view_t;    // per slot: 9, 9, 3
    std::vector<int32_t> view_valid;
    std::vector<int32_t> view_wh;                 // per slot: NMS box width, height (-1: the template's size)
    int32_t* d_view_wh = nullptr;
    bool views_dirty = true;
    float* d_view_K = nullptr;
    int32_t* d_view_valid = nullptr;
    int view_cap = 0;
    int32_t* d_class_base = nullptr;
    int class_cap = 0;
    TopkSel* d_sel = nullptr;
    int32_t* d_nsel = nullptr;
    int sel_cap = 0;
    void* d_scratch = nullptr;
    size_t scratch_cap = 0;
    TopkSel* h_sel = nullptr;                     // pinned
    int32_t* h_nsel = nullptr;                    // pinned
    int32_t* h_class_base = nullptr;              // pinned
    int h_cap = 0, h_class_cap = 0;
    std::vector<int32_t> class_base_on_device;    // what d_class_base holds (re-uploaded only when it changes)
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
};

extern "C" int lm_pipeline_create(lm_detector* det, int width, int height, lm_pipeline** out) {
    if (!det || !out || width <= 0 || height <= 0) return lm_set_error(LM_ERR_INVALID, "null argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(det->device));
    lm_pipeline* p = new lm_pipeline();
    p->det = det; p->W = width; p->H = height;
    int rc = lm_icp_create(det->device, &p->icp);
    if (!rc) rc = lm_icp_set_geometry(p->icp, width, height);
    if (rc) { if (p->icp) lm_icp_destroy(p->icp); delete p; return rc; }
    if (hipEventCreate(&p->e0) != hipSuccess || hipEventCreate(&p->e1) != hipSuccess || hipEventCreate(&p->e2) != hipSuccess) {
        lm_icp_destroy(p->icp); delete p;
        return lm_set_error(LM_ERR_HIP, "could not create the pipeline events");
    }
    *out = p;
    return LM_OK;
}

extern "C" void lm_pipeline_destroy(lm_pipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->det->device);
    (void)hipStreamSynchronize(p->det->stream);
    (void)hipStreamSynchronize(p->det->mstream);
    void* dev[] = {p->d_view_K, p->d_view_valid, p->d_view_wh, p->d_class_base, p->d_sel, p->d_nsel, p->d_scratch};
    for (void* q : dev) if (q) (void)hipFree(q);
    void* pin[] = {p->h_sel, p->h_nsel, p->h_class_base};
    for (void* q : pin) if (q) (void)hipHostFree(q);
    if (p->e0) (void)hipEventDestroy(p->e0);
    if (p->e1) (void)hipEventDestroy(p->e1);
    if (p->e2) (void)hipEventDestroy(p->e2);
    lm_icp_destroy(p->icp);
    delete p;
}

extern "C" int lm_pipeline_set_views(lm_pipeline* p, const char* class_id, int first_template, int count, const uint16_t* const* depth_ren,
                                     const float* Ks, const float* Rs, const float* ts, const int32_t* box_wh) {
    if (!p || !class_id || first_template < 0 || count < 0 || (count && (!depth_ren || !Ks || !Rs || !ts)))
        return lm_set_error(LM_ERR_INVALID, "null argument");
    const int nt = lm_detector_num_templates(p->det, class_id);
    if (nt <= 0) return lm_set_error(LM_ERR_NOT_FOUND, "class '%s' has no templates in the detector", class_id);
    if (first_template + count > nt) return lm_set_error(LM_ERR_INVALID, "views [%d, %d) exceed the %d templates of class '%s'", first_template, first_template + count, nt, class_id);
    auto it = p->views.find(class_id);
    if (it == p->views.end()) {
        lm_pipeline::ClassViews cv;
        cv.base = p->num_views; cv.count = nt;
        p->num_views += nt;
        p->view_K.resize((size_t)p->num_views * 9, 0.f); p->view_R.resize((size_t)p->num_views * 9, 0.f);
        p->view_t.resize((size_t)p->num_views * 3, 0.f); p->view_valid.resize((size_t)p->num_views, 0);
        p->view_wh.resize((size_t)p->num_views * 2, -1);
        it = p->views.emplace(class_id, cv).first;
    } else if (it->second.count != nt) {
        return lm_set_error(LM_ERR_INVALID, "class '%s' changed its template count after views were set", class_id);
    }
    if (count == 0) return LM_OK;
    HIP_TRY(hipSetDevice(p->det->device));
    HIP_TRY(hipStreamSynchronize(p->det->stream));                 // the slot array may be reallocated: no frame may be in flight
    HIP_TRY(hipStreamSynchronize(p->det->mstream));
    const int slot0 = it->second.base + first_template;
    int rc = lm_icp_set_models(p->icp, slot0, count, depth_ren);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->icp->s));
    memcpy(&p->view_K[(size_t)slot0 * 9], Ks, (size_t)count * 9 * sizeof(float));
    memcpy(&p->view_R[(size_t)slot0 * 9], Rs, (size_t)count * 9 * sizeof(float));
    memcpy(&p->view_t[(size_t)slot0 * 3], ts, (size_t)count * 3 * sizeof(float));
    for (int i = 0; i < count; ++i) {
        p->view_valid[(size_t)slot0 + i] = 1;
        p->view_wh[2 * ((size_t)slot0 + i)] = box_wh ? box_wh[2 * i] : -1;
        p->view_wh[2 * ((size_t)slot0 + i) + 1] = box_wh ? box_wh[2 * i + 1] : -1;
    }
    p->views_dirty = true;
    return LM_OK;
}

// Views rendered on the device: the depth of template first_template + i goes from the rasteriser straight into the
// ICP context's model slot — what the driver does per match at linemod_and_levelup_test.py:352, once per template.
extern "C" int lm_pipeline_set_views_rendered(lm_pipeline* p, lm_mesh* m, const char* class_id, int first_template, int count,
                                              const float* Ks, const float* Rs, const float* ts, float clip_near, float clip_far,
                                              const int32_t* box_wh) {
    if (!p || !m || !class_id || first_template < 0 || count < 0 || (count && (!Ks || !Rs || !ts)))
        return lm_set_error(LM_ERR_INVALID, "null argument");
    if (m->device != p->det->device) return lm_set_error(LM_ERR_INVALID, "mesh and detector live on different devices");
    const int nt = lm_detector_num_templates(p->det, class_id);
    if (nt <= 0) return lm_set_error(LM_ERR_NOT_FOUND, "class '%s' has no templates in the detector", class_id);
    if (first_template + count > nt) return lm_set_error(LM_ERR_INVALID, "views [%d, %d) exceed the %d templates of class '%s'", first_template, first_template + count, nt, class_id);
    auto it = p->views.find(class_id);
    if (it == p->views.end()) {
        lm_pipeline::ClassViews cv;
        cv.base = p->num_views; cv.count = nt;
        p->num_views += nt;
        p->view_K.resize((size_t)p->num_views * 9, 0.f); p->view_R.resize((size_t)p->num_views * 9, 0.f);
        p->view_t.resize((size_t)p->num_views * 3, 0.f); p->view_valid.resize((size_t)p->num_views, 0);
        p->view_wh.resize((size_t)p->num_views * 2, -1);
        it = p->views.emplace(class_id, cv).first;
    } else if (it->second.count != nt) {
        return lm_set_error(LM_ERR_INVALID, "class '%s' changed its template count after views were set", class_id);
    }
    if (count == 0) return LM_OK;
    HIP_TRY(hipSetDevice(p->det->device));
    HIP_TRY(hipStreamSynchronize(p->det->stream));
    HIP_TRY(hipStreamSynchronize(p->det->mstream));
    const int slot0 = it->second.base + first_template;
    int rc = lm_icp_ensure_slots(p->icp, slot0 + count);
    if (rc) return rc;
    const size_t npx = (size_t)p->W * p->H;
    const int chunk = 256;
    for (int c0 = 0; c0 < count; c0 += chunk) {
        const int n = std::min(chunk, count - c0);
        if ((rc = lm_mesh_render_device(m, n, p->W, p->H, Ks + 9 * (size_t)c0, Rs + 9 * (size_t)c0, ts + 3 * (size_t)c0, clip_near, clip_far, 0.f,
                                        1, true, false)))
            return rc;
        HIP_TRY(hipMemcpyAsync(p->icp->d_models + ((size_t)slot0 + c0) * npx, m->d_depth, (size_t)n * npx * sizeof(uint16_t),
                               hipMemcpyDeviceToDevice, m->s));
        lm::launch_icp_model_boxes(p->icp->d_models, p->icp->d_model_bbox, slot0 + c0, n, p->W, p->H, m->s);   // the boxes of the new views, once (LL.cpp:43-50)
        for (int i = 0; i < n; ++i) p->icp->slot_boxed[(size_t)slot0 + c0 + i] = 1;
        HIP_TRY(hipStreamSynchronize(m->s));
    }
    memcpy(&p->view_K[(size_t)slot0 * 9], Ks, (size_t)count * 9 * sizeof(float));
    memcpy(&p->view_R[(size_t)slot0 * 9], Rs, (size_t)count * 9 * sizeof(float));
    memcpy(&p->view_t[(size_t)slot0 * 3], ts, (size_t)count * 3 * sizeof(float));
    for (int i = 0; i < count; ++i) {
        p->view_valid[(size_t)slot0 + i] = 1;
        p->view_wh[2 * ((size_t)slot0 + i)] = box_wh ? box_wh[2 * i] : -1;
        p->view_wh[2 * ((size_t)slot0 + i) + 1] = box_wh ? box_wh[2 * i + 1] : -1;
    }
    p->views_dirty = true;
    return LM_OK;
}

static int ensure_run_buffers(lm_pipeline* p, int top_k, int num_classes) {
    lm_detector* d = p->det;
    if (p->views_dirty || p->view_cap < p->num_views) {
        if (p->view_cap < p->num_views) {
            if (p->d_view_K) (void)hipFree(p->d_view_K);
            if (p->d_view_valid) (void)hipFree(p->d_view_valid);
            if (p->d_view_wh) (void)hipFree(p->d_view_wh);
            p->d_view_K = nullptr; p->d_view_valid = nullptr; p->d_view_wh = nullptr;
            p->view_cap = std::max(p->num_views, 1);
            HIP_TRY(hipMalloc((void**)&p->d_view_K, (size_t)p->view_cap * 9 * sizeof(float)));
            HIP_TRY(hipMalloc((void**)&p->d_view_valid, (size_t)p->view_cap * sizeof(int32_t)));
            HIP_TRY(hipMalloc((void**)&p->d_view_wh, (size_t)p->view_cap * 2 * sizeof(int32_t)));
        }
        if (p->num_views) {
            HIP_TRY(hipMemcpy(p->d_view_K, p->view_K.data(), (size_t)p->num_views * 9 * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_view_valid, p->view_valid.data(), (size_t)p->num_views * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_view_wh, p->view_wh.data(), (size_t)p->num_views * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        p->views_dirty = false;
    }
    if (top_k > p->sel_cap) {
        if (p->d_sel) (void)hipFree(p->d_sel);
        if (p->h_sel) (void)hipHostFree(p->h_sel);
        p->d_sel = nullptr; p->h_sel = nullptr;
        HIP_TRY(hipMalloc((void**)&p->d_sel, (size_t)top_k * sizeof(TopkSel)));
        HIP_TRY(hipHostMalloc((void**)&p->h_sel, (size_t)top_k * sizeof(TopkSel), hipHostMallocDefault));
        p->sel_cap = top_k;
    }
    if (!p->d_nsel) {
        HIP_TRY(hipMalloc((void**)&p->d_nsel, 2 * sizeof(int32_t)));
        HIP_TRY(hipHostMalloc((void**)&p->h_nsel, 2 * sizeof(int32_t), hipHostMallocDefault));
    }
    if (num_classes > p->class_cap) {
        if (p->d_class_base) (void)hipFree(p->d_class_base);
        if (p->h_class_base) (void)hipHostFree(p->h_class_base);
        p->d_class_base = nullptr; p->h_class_base = nullptr;
        p->class_base_on_device.clear();
        p->class_cap = std::max(num_classes, 8);
        HIP_TRY(hipMalloc((void**)&p->d_class_base, (size_t)p->class_cap * sizeof(int32_t)));
        HIP_TRY(hipHostMalloc((void**)&p->h_class_base, (size_t)p->class_cap * sizeof(int32_t), hipHostMallocDefault));
    }
    const size_t need = topk_nms_scratch_bytes(d->cand_cap);
    if (need > p->scratch_cap) {
        if (p->d_scratch) (void)hipFree(p->d_scratch);
        p->d_scratch = nullptr; p->scratch_cap = 0;
        HIP_TRY(hipMalloc(&p->d_scratch, need));
        p->scratch_cap = need;
    }
    return LM_OK;
}

extern "C" int lm_pipeline_run(lm_pipeline* p, float threshold, const char* const* class_ids, int num_class_ids, const float* scene_K,
                               int top_k, double nms_iou, int flags, lm_detection* out, int* n_out, lm_pipeline_timings* tm) {
    if (!p || !scene_K || top_k <= 0 || !out || !n_out) return lm_set_error(LM_ERR_INVALID, "null argument");
    *n_out = 0;
    lm_detector* d = p->det;
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "a frame is in flight on the detector: collect it first");
    if (!d->frame_valid || d->fW != p->W || d->fH != p->H)
        return lm_set_error(LM_ERR_INVALID, "the detector's resident frame is not %dx%d", p->W, p->H);
    HIP_TRY(hipSetDevice(d->device));
    hipStream_t s = d->mstream;                                   // after the matching kernels of the frame
    lm_icp* c = p->icp;
    int rc;
    if ((rc = lm_icp_ensure_arenas(c, top_k))) return rc;
    int overflows = 0;
    for (;;) {                                                    // one pass normally; again after a candidate-buffer overflow (<= 3) 
        HIP_TRY(hipEventRecord(p->e0, d->stream));
        if ((rc = lm_submit_frame(d, threshold, class_ids, num_class_ids))) return rc;
        // class position (caller's class_ids order, or sorted order) -> first view slot
        std::vector<std::string> order;
        if (class_ids && num_class_ids > 0) for (int i = 0; i < num_class_ids; ++i) order.push_back(class_ids[i] ? class_ids[i] : "");
        else order = d->bank_classes;
        if ((rc = ensure_run_buffers(p, top_k, (int)order.size()))) return rc;
        for (size_t i = 0; i < order.size(); ++i) {
            auto it = p->views.find(order[i]);
            p->h_class_base[i] = it == p->views.end() ? -1 : it->second.base;
        }
        if (!order.empty() && (p->class_base_on_device.size() != order.size() ||
                               memcmp(p->class_base_on_device.data(), p->h_class_base, order.size() * sizeof(int32_t)) != 0)) {
            // the stream order keeps earlier frames' kernels ahead of this copy; the pinned source is stable until the next change
            HIP_TRY(hipMemcpyAsync(p->d_class_base, p->h_class_base, order.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            p->class_base_on_device.assign(p->h_class_base, p->h_class_base + order.size());
        }
        HIP_TRY(hipEventRecord(p->e1, s));
        const int slot = (int)((d->n_submitted - 1) % lm_detector::kSlots);          // the frame just submitted
        launch_topk_nms(d->d_matches_dev.p + (size_t)d->buf_cand_cap * slot, d->d_final.p + 8 * (size_t)slot, d->buf_cand_cap, d->d_work.p, d->d_work_cls.p, d->d_work_tid.p, d->d_entries.p,
                        d->pyramid_levels, p->d_class_base, p->d_view_wh, p->num_views, top_k, nms_iou, p->d_scratch, p->d_sel, p->d_nsel, s);
        launch_icp_bind(p->d_sel, p->d_nsel, p->d_class_base, p->d_view_K, p->d_view_valid, p->num_views, c->d_in, c->d_st, top_k, s);
        HIP_TRY(hipEventRecord(p->e2, s));
        IcpBuffers B = c->B;
        B.scene = d->cur_depth; B.models = c->d_models; B.model_bbox = c->d_model_bbox; B.in = c->d_in; B.st = c->d_st;
        B.count = top_k;
        memcpy(B.sK, scene_K, sizeof(B.sK));
        // (k_icp_bind only binds views that were uploaded, and both upload paths work out the boxes: 0x100 = no k_icp_bbox)
        launch_icp_pipeline(B, top_k, p->W, p->H, (flags & 0xFF) | 0x100, kVoxel, kMaxDist, kMaxIter, kRelTol, kKnn, c->solo_from, s);
        HIP_TRY(hipEventRecord(c->e1, s));
        HIP_TRY(hipMemcpyAsync(c->h_st, c->d_st, (size_t)top_k * sizeof(IcpState), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(p->h_sel, p->d_sel, (size_t)top_k * sizeof(TopkSel), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(p->h_nsel, p->d_nsel, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipGetLastError());
        rc = lm_collect_frame(d, -1, nullptr, nullptr);          // retires the frame; 1 = candidate buffer overflow, rerun
        if (rc == 1) {
            if (++overflows >= 3) return lm_set_error(LM_ERR_INVALID, "candidate buffer kept overflowing");
            continue;
        }
        if (rc) return rc;
        for (int pass = 0; pass < 2 && lm_icp_unfinished(c->h_st, top_k); ++pass) {
            // clouds the first team builds do not hold (more than 704 source points per workgroup): the builds with more points per thread;
            // what those leave too, or a team that timed out: the sliced launches
            if (c->solo_from != 0) return lm_set_error(LM_ERR_HIP, "ICP: a hypothesis was left unfinished");
            if (pass == 0) launch_icp_team(B, top_k, 1, kMaxDist, kMaxIter, kRelTol, s);
            else launch_icp_evals(B, top_k, 0, kMaxIter + 1, kMaxDist, kMaxIter, kRelTol, s);
            HIP_TRY(hipEventRecord(c->e1, s));
            HIP_TRY(hipMemcpyAsync(c->h_st, c->d_st, (size_t)top_k * sizeof(IcpState), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipGetLastError());
        }
        if (lm_icp_unfinished(c->h_st, top_k)) return lm_set_error(LM_ERR_HIP, "ICP: a hypothesis was left unfinished");
        break;
    }
    if (p->h_nsel[1] != 0)
        return lm_set_error(LM_ERR_INVALID, "on-device NMS: a match field exceeds the packed record (template id >= 2^24, class position >= 128 or |x|,|y| >= 32768)");
    c->last_count = top_k; c->last_flags = flags;
    const int n = p->h_nsel[0];
    for (int i = 0; i < n; ++i) {
        const TopkSel& sl = p->h_sel[i];
        const IcpState& st = c->h_st[i];
        lm_detection& o = out[i];
        memset(&o, 0, sizeof(o));
        o.match.x = sl.x; o.match.y = sl.y; o.match.similarity = sl.similarity;
        o.match.class_index = sl.class_index; o.match.template_id = sl.template_id;
        o.width = sl.width; o.height = sl.height;
        o.pose.residual = -1.f;
        o.status = st.status;
        if (st.status == 2) return lm_set_error(LM_ERR_INVALID, "rendered depth of template %d is empty", sl.template_id);
        if (st.status == 3) return lm_set_error(LM_ERR_INVALID, "detection %d: point cloud too large for 64-bit voxel keys", i);
        if (st.status == lm::kIcpStalled) return lm_set_error(LM_ERR_HIP, "detection %d: the point kernels stalled (strips waited a second for each other)", i);
        if (st.status != 0) continue;                             // 1: window leaves the frame (LL.cpp:52-55); 5: no view for the template
        const int base = p->h_class_base[sl.class_index];
        const size_t v = (size_t)base + sl.template_id;
        lm_icp_compose_result(st, &p->view_R[v * 9], &p->view_t[v * 3], &o.pose);
    }
    *n_out = n;
    if (tm) {
        memset(tm, 0, sizeof(*tm));
        (void)hipEventElapsedTime(&tm->match_ms, p->e0, p->e1);
        (void)hipEventElapsedTime(&tm->nms_ms, p->e1, p->e2);
        (void)hipEventElapsedTime(&tm->icp_ms, p->e2, c->e1);
        (void)hipEventElapsedTime(&tm->total_ms, p->e0, c->e1);
        tm->coarse_candidates = d->timings.coarse_candidates;
        tm->matches_pre_unique = d->timings.matches_pre_unique;
        int its = 0;
        for (int i = 0; i < n; ++i) if (c->h_st[i].status == 0) its += c->h_st[i].iterations;
        tm->icp_iterations = its;
    }
    return LM_OK;
}

extern "C" int64_t lm_pipeline_read_icp_debug(lm_pipeline* p, int hypothesis, int kind, double* dst, int64_t capacity) {
    if (!p) return lm_set_error(LM_ERR_INVALID, "null argument");
    return lm_icp_read_debug(p->icp, hypothesis, kind, dst, capacity);
}
