// ICP inner loops of poseRefine::process on gfx950 (reference call sites LL.cpp:127-130; the
// arithmetic is Open3D's — EstimateNormals(KNN 30), RegistrationICP with
// TransformationEstimationPointToPlane — restated per SURVEY Appendix B).
//
// One workgroup per pose hypothesis; the whole ICP (<= 30 iterations: nearest-neighbour
// correspondence + 6x6 point-to-plane normal equations + solve + transform + convergence test)
// runs inside ONE launch.  Clouds are ~1.3k points, so brute-force NN with the target streamed
// through the scalar cache (the inner index is wave-uniform) beats any tree; all arithmetic is
// double like Open3D's (f64 VALU, no MFMA: nothing here is a dense contraction).  The 29 partial
// sums are reduced with wave shuffles, then across the 4 waves through LDS.
#include "icp_kernels.h"

namespace lm {

static __device__ __forceinline__ double sqdist(double ax, double ay, double az, double bx, double by, double bz) {
    double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// ---- 3x3 symmetric eigen decomposition (cyclic Jacobi), eigenvector of the smallest eigenvalue ----
static __device__ void smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double n[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[p][q];
                if (apq == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {   // A <- A * G
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {   // A <- G^T * A
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    n[0] = V[0][m]; n[1] = V[1][m]; n[2] = V[2][m];
}

// ---- EstimateNormals(KNN): one thread per target point; k passes, each selecting the next
// neighbour in (distance, index) order — no per-thread arrays, target streamed via scalar loads ----
__global__ void __launch_bounds__(256)
k_knn_normals(const double* __restrict__ pts, double* __restrict__ normals, const IcpProblem* __restrict__ probs, int knn) {
    const IcpProblem pb = probs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= pb.n_tgt) return;
    const bool active = i < pb.n_tgt;
    const double* T = pts + 3 * (size_t)pb.tgt_off;
    const int ii = active ? i : 0;
    const double px = T[3 * ii], py = T[3 * ii + 1], pz = T[3 * ii + 2];
    const int n = pb.n_tgt;
    const int k = knn < n ? knn : n;
    double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    double prev_d = -1.0;
    int prev_j = -1;
    for (int pass = 0; pass < k; ++pass) {
        double bd = 1e300;
        int bj = -1;
        for (int j = 0; j < n; ++j) {
            double qx = T[3 * j], qy = T[3 * j + 1], qz = T[3 * j + 2];
            double d = sqdist(px, py, pz, qx, qy, qz);
            bool after = d > prev_d || (d == prev_d && j > prev_j);
            if (after && d < bd) { bd = d; bj = j; }       // strict '<': lowest index wins ties
        }
        prev_d = bd; prev_j = bj;
        if (bj < 0) break;
        double qx = T[3 * bj], qy = T[3 * bj + 1], qz = T[3 * bj + 2];
        sx += qx; sy += qy; sz += qz;
        sxx += qx * qx; sxy += qx * qy; sxz += qx * qz; syy += qy * qy; syz += qy * qz; szz += qz * qz;
    }
    double nrm[3] = {0, 0, 1};
    if (k >= 3) {
        double mx = sx / k, my = sy / k, mz = sz / k;
        smallest_eigvec(sxx / k - mx * mx, sxy / k - mx * my, sxz / k - mx * mz, syy / k - my * my, syz / k - my * mz,
                        szz / k - mz * mz, nrm);
        if (nrm[0] == 0 && nrm[1] == 0 && nrm[2] == 0) nrm[2] = 1;
    }
    if (active) {
        double* o = normals + 3 * ((size_t)pb.tgt_off + i);
        o[0] = nrm[0]; o[1] = nrm[1]; o[2] = nrm[2];
    }
}

void launch_knn_normals(const double* pts, double* normals, const IcpProblem* probs, int count, int max_tgt, int knn,
                        hipStream_t s) {
    if (count <= 0 || max_tgt <= 0) return;
    hipLaunchKernelGGL(k_knn_normals, dim3((max_tgt + 255) / 256, count), dim3(256), 0, s, pts, normals, probs, knn);
}

// ---- RegistrationICP ------------------------------------------------------------------------------
constexpr int kNSum = 29;   // 21 JtJ (upper) + 6 Jtr + sum d^2 + count

static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Gaussian elimination with partial pivoting, A x = b (6x6).  Returns false if singular / non-finite.
static __device__ bool solve6(double A[6][6], double b[6], double x[6]) {
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < 6; ++r)
            if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
        if (!(best > 0.0)) return false;
        if (piv != c) {
            for (int k = 0; k < 6; ++k) { double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
            double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        for (int r = c + 1; r < 6; ++r) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 6; ++k) A[r][k] -= f * A[c][k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < 6; ++k) s -= A[r][k] * x[k];
        x[r] = s / A[r][r];
    }
    for (int r = 0; r < 6; ++r)
        if (!isfinite(x[r])) return false;
    return true;
}

__global__ void __launch_bounds__(256)
k_icp(const double* __restrict__ pts, const double* __restrict__ normals, double* __restrict__ work /* transformed src */,
      const IcpProblem* __restrict__ probs, IcpResult* __restrict__ results, double max_dist, int max_iter, double rel_tol) {
    __shared__ double s_part[4][kNSum];
    __shared__ double s_sum[kNSum];
    __shared__ double s_upd[12];     // 3x4 update
    __shared__ int s_stop;
    __shared__ double s_T[16];
    __shared__ double s_fit, s_rmse;
    __shared__ int s_iters, s_ncorr;

    const IcpProblem pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* S = pts + 3 * (size_t)pb.src_off;
    const double* T = pts + 3 * (size_t)pb.tgt_off;
    const double* N = normals + 3 * (size_t)pb.tgt_off;
    double* P = work + 3 * (size_t)pb.src_off;
    const int ns = pb.n_src, nt = pb.n_tgt;
    const double r2 = max_dist * max_dist;

    if (tid < 16) s_T[tid] = pb.init[tid];
    if (tid == 0) { s_stop = 0; s_iters = 0; s_fit = 0; s_rmse = 0; s_ncorr = 0; }
    __syncthreads();
    // pcd.Transform(init)
    for (int i = tid; i < ns; i += blockDim.x) {
        double x = S[3 * i], y = S[3 * i + 1], z = S[3 * i + 2];
        P[3 * i] = s_T[0] * x + s_T[1] * y + s_T[2] * z + s_T[3];
        P[3 * i + 1] = s_T[4] * x + s_T[5] * y + s_T[6] * z + s_T[7];
        P[3 * i + 2] = s_T[8] * x + s_T[9] * y + s_T[10] * z + s_T[11];
    }
    __syncthreads();

    for (int it = 0; it <= max_iter; ++it) {
        // --- GetRegistrationResultAndCorrespondences fused with the JtJ / Jtr accumulation ---
        double acc[kNSum];
#pragma unroll
        for (int k = 0; k < kNSum; ++k) acc[k] = 0.0;
        for (int i0 = 0; i0 < ns; i0 += blockDim.x) {
            const int i = i0 + tid;
            const bool act = i < ns;
            const int ii = act ? i : 0;
            const double px = P[3 * ii], py = P[3 * ii + 1], pz = P[3 * ii + 2];
            double bd = 1e300;
            int bj = -1;
            for (int j = 0; j < nt; ++j) {                       // wave-uniform index: scalar loads
                double d = sqdist(px, py, pz, T[3 * j], T[3 * j + 1], T[3 * j + 2]);
                if (d < bd) { bd = d; bj = j; }
            }
            if (act && bj >= 0 && bd < r2) {
                const double qx = T[3 * bj], qy = T[3 * bj + 1], qz = T[3 * bj + 2];
                const double nx = N[3 * bj], ny = N[3 * bj + 1], nz = N[3 * bj + 2];
                const double r = (px - qx) * nx + (py - qy) * ny + (pz - qz) * nz;
                double J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = a; b < 6; ++b) acc[k++] += J[a] * J[b];
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
                acc[27] += bd;
                acc[28] += 1.0;
            }
        }
#pragma unroll
        for (int k = 0; k < kNSum; ++k) {
            double v = wave_sum(acc[k]);
            if (lane == 0) s_part[wave][k] = v;
        }
        __syncthreads();
        if (tid < kNSum) {
            double v = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_part[w][tid];
            s_sum[tid] = v;
        }
        __syncthreads();
        if (tid == 0) {
            const int ncorr = (int)s_sum[28];
            const double fit = ncorr ? (double)ncorr / (double)ns : 0.0;
            const double rmse = ncorr ? sqrt(s_sum[27] / (double)ncorr) : 0.0;
            if (it > 0 && fabs(s_fit - fit) < rel_tol && fabs(s_rmse - rmse) < rel_tol) s_stop = 1;
            s_fit = fit; s_rmse = rmse; s_ncorr = ncorr;
            if (it == max_iter) s_stop = 1;
            if (!s_stop) {
                // TransformationEstimationPointToPlane::ComputeTransformation
                double A[6][6], b[6], x[6];
                int k = 0;
                for (int a = 0; a < 6; ++a)
                    for (int c = a; c < 6; ++c) { A[a][c] = s_sum[k]; A[c][a] = s_sum[k]; ++k; }
                for (int a = 0; a < 6; ++a) b[a] = -s_sum[21 + a];
                double U[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                if (ncorr >= 6 && solve6(A, b, x)) {
                    const double cx = cos(x[0]), sx = sin(x[0]), cy = cos(x[1]), sy = sin(x[1]), cz = cos(x[2]), sz = sin(x[2]);
                    // Rz(x2) * Ry(x1) * Rx(x0)
                    U[0] = cz * cy; U[1] = cz * sy * sx - sz * cx; U[2] = cz * sy * cx + sz * sx; U[3] = x[3];
                    U[4] = sz * cy; U[5] = sz * sy * sx + cz * cx; U[6] = sz * sy * cx - cz * sx; U[7] = x[4];
                    U[8] = -sy;     U[9] = cy * sx;                U[10] = cy * cx;               U[11] = x[5];
                }
                for (int a = 0; a < 12; ++a) s_upd[a] = U[a];
                // transformation = update * transformation
                double Tn[12];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c)
                        Tn[4 * r + c] = U[4 * r] * s_T[c] + U[4 * r + 1] * s_T[4 + c] + U[4 * r + 2] * s_T[8 + c] + (c == 3 ? U[4 * r + 3] : 0.0);
                for (int a = 0; a < 12; ++a) s_T[a] = Tn[a];
                s_iters = it + 1;
            }
        }
        __syncthreads();
        if (s_stop) break;
        // pcd.Transform(update)
        for (int i = tid; i < ns; i += blockDim.x) {
            double x = P[3 * i], y = P[3 * i + 1], z = P[3 * i + 2];
            P[3 * i] = s_upd[0] * x + s_upd[1] * y + s_upd[2] * z + s_upd[3];
            P[3 * i + 1] = s_upd[4] * x + s_upd[5] * y + s_upd[6] * z + s_upd[7];
            P[3 * i + 2] = s_upd[8] * x + s_upd[9] * y + s_upd[10] * z + s_upd[11];
        }
        __syncthreads();
    }
    if (tid == 0) {
        IcpResult& r = results[blockIdx.x];
        for (int a = 0; a < 16; ++a) r.T[a] = s_T[a];
        r.fitness = s_fit; r.rmse = s_rmse; r.iterations = s_iters; r.n_corr = s_ncorr;
    }
}

void launch_icp(const double* pts, const double* normals, double* work, const IcpProblem* probs, IcpResult* results,
                int count, double max_dist, int max_iter, double rel_tol, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_icp, dim3(count), dim3(256), 0, s, pts, normals, work, probs, results, max_dist, max_iter, rel_tol);
}

}  // namespace lm
