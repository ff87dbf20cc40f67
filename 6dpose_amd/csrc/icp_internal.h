// Private view of the ICP context shared by pose_refine.cpp and pipeline.cpp (not part of the C ABI).
#pragma once
#include "../../include/amd_linemod.h"
#include "icp_kernels.h"
#include <vector>

using lm::IcpBuffers;
using lm::IcpIn;
using lm::IcpState;

struct lm_icp {
    int device = 0;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int W = 0, H = 0;
    bool have_scene = false;
    float sK[9] = {0};
    int slots = 0;                 // resident model depth images
    int max_count = 0;             // hypotheses the arenas hold
    int last_count = 0, last_flags = 0;
    uint16_t* d_scene = nullptr;
    uint16_t* d_models = nullptr;
    int* d_model_bbox = nullptr;   // [slots][8] bounding box of a resident model depth image (x0, y0, x1, y1, state: 1 = known, -, -, -): worked out at upload (k_icp_model_boxes)
    std::vector<uint8_t> slot_boxed;   // host view of the state words: 1 = the slot's box was worked out at upload
    IcpIn* d_in = nullptr;
    IcpState* d_st = nullptr;
    IcpBuffers B{};
    void* pinned = nullptr;        // staging for images
    size_t pinned_bytes = 0;
    IcpIn* h_in = nullptr;         // pinned
    IcpState* h_st = nullptr;      // pinned: the states uploaded before / read after a run
    IcpState* h_st2 = nullptr;     // pinned: read-back of lm_icp_run (h_st keeps the initial states for a repeated run)
    int h_cap = 0;
    int solo_from = 0;             // 0: RegistrationICP as one launch (k_icp_team), the sliced launches only for hypotheses it leaves unfinished;
                                   // -1 (LM_ICP_SLICED=1): one launch per evaluation (k_icp_eval, rounds 1-5)
};


// pose_refine.cpp
bool lm_icp_unfinished(const lm::IcpState* st, int count);   // a hypothesis k_icp_solo left to the sliced launches (status 0, stop 0)
int lm_icp_set_geometry(lm_icp* c, int W, int H);      // (re)allocates for a frame size; drops the slots when it changes
int lm_icp_ensure_arenas(lm_icp* c, int count);        // arenas for `count` hypotheses
int lm_icp_ensure_slots(lm_icp* c, int slots);         // resident model depth images
// [R|t] of LL.cpp:34-41 and LL.cpp:146-154 from the device state of one hypothesis
void lm_icp_compose_result(const lm::IcpState& st, const float* model_R, const float* model_t, lm_pose_result* o);
