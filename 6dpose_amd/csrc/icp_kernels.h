// Internal interface between pose_refine.cpp (host) and icp.hip (kernels).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lm {

struct IcpProblem {          // one pose hypothesis; clouds are xyz triples of doubles in one arena
    int src_off, n_src;      // source cloud (voxel-down-sampled model), point index into the arena
    int tgt_off, n_tgt;      // target cloud
    double init[16];         // row-major 4x4 initial guess
};
struct IcpResult {
    double T[16];            // final transformation_ (row-major)
    double fitness, rmse;    // fitness_, inlier_rmse_
    int iterations, n_corr;
};

void launch_knn_normals(const double* pts, double* normals, const IcpProblem* probs, int count, int max_tgt, int knn,
                        hipStream_t s);
void launch_icp(const double* pts, const double* normals, double* work, const IcpProblem* probs, IcpResult* results,
                int count, double max_dist, int max_iter, double rel_tol, hipStream_t s);

}  // namespace lm
