// Tuning knobs of the library, read from the environment ONCE (first use = creation of the first detector / ICP context), never
// on a launch path.  Every knob here leaves the results unchanged.  The knobs of the timing experiments under profiles/ that
// produce WRONG results (skipping a phase of a kernel, stopping ICP early) exist only in a build with -DLM_DIAG
// (`make DIAG=1`); the default library does not read them.
#pragma once
#include <stdlib.h>

#ifndef LM_FE_WAVES
#define LM_FE_WAVES 7
#endif
constexpr int kFeWaves = LM_FE_WAVES;   // waves per SIMD = persistent workgroups per CU of k_fe_stage: 7 = 72 VGPRs, no spills (6: 78 VGPRs; 8: 64 VGPRs with 7 spilled).  k_fe_stage per launch, mean of the two of an 8-frame batch: 50.8 / 48.6 / 47.0 us at 6 / 7 / 8 (profiles/r04_stream_ab.txt)

namespace lm {

struct Knobs {
    int coarse_group = 4;          // LM_COARSE_GROUP: templates per workgroup of k_coarse (<= 4)
    int local_blocks = 0;          // LM_LOCAL_BLOCKS: grid of k_local (0 = default per CU count)
    int frame_batch = 0;           // LM_FRAME_BATCH: frames per matching launch in stream mode (0 = default: 8 = kMaxBatch; lm_detector_set_batch)
    int coarse_bits = 1;           // LM_COARSE_BITS=0: coarse pass on the byte linear memories (k_coarse) instead of on the pair stream (k_coarse_bits)
    int bitplanes = 1;             // LM_BITPLANES=0: refinement on the byte strip planes with tiles (round 2-3's kernel) instead of on bit planes
    int fe_bits = 1;               // LM_FE_BITS=0: the front end always writes the byte planes and k_pack_bits / k_pack_top pack them (1: bit planes directly when nothing reads the bytes)
    int fe_bits_split = 0;         // LM_FE_BITS_SPLIT=1: k_fe_bits as two launches (strip records, pair stream) so that a profile times them apart
    int first_batch = 3;           // LM_FIRST_BATCH: frames an idle GPU waits for before a partial batch goes out WHILE THE CALLER SUBMITS IN A TIGHT LOOP (collect / flush launch what is left; sparse streams: every frame at once)
    int fe_rows_cs = 0;            // LM_FE_ROWS_CS: column phases per workgroup of the strip-record tile writer (a divisor of T: 1, 2, 4, 8 or 1, 5; 0 = default 2, or 1 where 2 does not divide T: the spread rows are built T / 2 times per row phase, for twice the workgroups)
    int nt_copy = 1;               // LM_NT_COPY=0: the staging copy of a streamed frame with memcpy instead of non-temporal AVX2 stores
    int dedupe_blocks = 0;         // LM_DEDUPE_BLOCKS: workgroups per frame of k_dedupe (0 = default: one per 256 candidates of the last frame, 64 .. two per CU)
    int fe_wgs_per_cu = kFeWaves;         // LM_FE_WGS_PER_CU: persistent workgroups per CU of a front-end stage (0 = one workgroup per tile)
    int launch_slack_us = 0;       // LM_LAUNCH_SLACK_US: a partial batch goes out when the GPU's estimated backlog is shorter than this (0 = default 150)
    int batch_queue = 0;           // LM_BATCH_QUEUE: launched batches to keep queued on the GPU before streamed frames wait for a full batch (0 = default: 2; lm_detector_set_batch_queue)
    int knn_blocks = 0;            // LM_KNN_BLOCKS: grid.x of k_icp_knn (0 = default: 64 workgroups per cloud up to 32 clouds, 32 beyond)
    int icp_splits = 0;            // LM_ICP_SPLITS: slices per hypothesis of k_icp_eval (0 = default schedule)
    int icp_team = 0;              // LM_ICP_TEAM: workgroups per hypothesis of k_icp_team (0 = default: 16, at most CUs / hypotheses)
    int icp_builds = 0;            // LM_ICP_BUILDS: which builds of k_icp_team are launched, in this order (bits: 1 = one point per thread, whole cloud only; 2 = one point, slab;
                                   // 4 = two points, slab; 8 = five points, slab; 0 = default 1 | 2 | 4, and 8 for batches whose teams are under four workgroups)
    int icp_team_min_points = 128; // LM_ICP_TEAM_MIN_POINTS: a hypothesis is never dealt more than one workgroup per this many source points
    int icp_cut_index = 3;         // LM_ICP_CUT_INDEX: the evaluation index after which a cramped batch leaves k_icp_team's first launch
    int icp_relaunch = 1;          // LM_ICP_RELAUNCH=0: k_icp_team as one launch; n: up to n more launches for hypotheses the teams suspend to have the chip dealt out again
    int icp_wide_sort = 1;         // LM_ICP_WIDE_SORT=0: voxel down-sampling and the search grid by one workgroup per cloud (k_icp_voxel / k_icp_grid alone), as before round 6
    int icp_maxshift = 3;          // LM_ICP_MAXSHIFT / _LATE: log2 lanes per searching point, early / late evaluations
    int icp_maxshift_late = 4;
#ifdef LM_DIAG
    int coarse_dbg = 0;            // LM_COARSE_DBG: 1 = no tile grouping, 2 = no global atomic (wrong results)
    int local_dbg = 0;             // LM_LOCAL_DBG: 1 = tiles only, 2 = singles only (wrong results)
    int icp_maxiter_diag = -1;     // LM_ICP_MAXITER_DIAG: stop ICP after this many evaluations (wrong results)
#endif
};

inline const Knobs& knobs() {
    static const Knobs k = [] {
        Knobs v;
        auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e && e[0] ? atoi(e) : dflt; };
        v.coarse_group = geti("LM_COARSE_GROUP", v.coarse_group);
        v.local_blocks = geti("LM_LOCAL_BLOCKS", 0);
        v.frame_batch = geti("LM_FRAME_BATCH", 0);
        v.batch_queue = geti("LM_BATCH_QUEUE", 0);
        v.launch_slack_us = geti("LM_LAUNCH_SLACK_US", 0);
        v.fe_wgs_per_cu = geti("LM_FE_WGS_PER_CU", kFeWaves);
        v.bitplanes = geti("LM_BITPLANES", 1);
        v.coarse_bits = geti("LM_COARSE_BITS", 1);
        v.fe_bits = geti("LM_FE_BITS", 1);
        v.fe_bits_split = geti("LM_FE_BITS_SPLIT", 0);
        v.first_batch = geti("LM_FIRST_BATCH", v.first_batch);
        v.dedupe_blocks = geti("LM_DEDUPE_BLOCKS", 0);
        v.nt_copy = geti("LM_NT_COPY", 1);
        v.fe_rows_cs = geti("LM_FE_ROWS_CS", 0);
        v.knn_blocks = geti("LM_KNN_BLOCKS", 0);
        v.icp_splits = geti("LM_ICP_SPLITS", 0);
        v.icp_team = geti("LM_ICP_TEAM", 0);
        v.icp_builds = geti("LM_ICP_BUILDS", 0);
        v.icp_wide_sort = geti("LM_ICP_WIDE_SORT", 1);
        v.icp_relaunch = geti("LM_ICP_RELAUNCH", 1);
        v.icp_cut_index = geti("LM_ICP_CUT_INDEX", 3);
        v.icp_team_min_points = geti("LM_ICP_TEAM_MIN_POINTS", 128);
        if (v.icp_team_min_points < 1) v.icp_team_min_points = 1;
        v.icp_maxshift = geti("LM_ICP_MAXSHIFT", v.icp_maxshift);
        v.icp_maxshift_late = geti("LM_ICP_MAXSHIFT_LATE", v.icp_maxshift_late);
#ifdef LM_DIAG
        v.coarse_dbg = geti("LM_COARSE_DBG", 0);
        v.local_dbg = geti("LM_LOCAL_DBG", 0);
        v.icp_maxiter_diag = geti("LM_ICP_MAXITER_DIAG", -1);
#endif
        return v;
    }();
    return k;
}

}  // namespace lm
