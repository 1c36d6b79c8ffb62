// Weighted least squares on the device: the M-step of the linear-regression path.
//
// Replaces  theta = lstsq(diag(sqrt(w)) @ X, diag(sqrt(w)) @ y)  (standard-learning/rlvi.py:70-71,
// :79-80; the reference materialises the n x n diagonal and calls scipy's LAPACK gelsd) by the
// normal equations  (Xa^T W Xa) with Xa = [X | y]:  the (d+1) x (d+1) weighted Gram matrix is THE
// dense contraction of this path (n*d^2 flops) and runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64); a Cholesky factorisation + two triangular solves of the d x d system
// finish in one workgroup.  Needs full column rank (the reference's lstsq returns the minimum-norm
// solution for rank-deficient X; here a non-positive pivot sets RLVI_ST_SINGULAR and theta = NaN).
//
// f64 MFMA operand maps (cdna_hip_programming.md section 3): lane l feeds A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; the four results of a lane are D[row = (l>>4) + 4*reg][col = l&15].
#include "rlvi_common.h"

namespace rlvi {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int WLS_MAXD = 63;                 // columns of X; [X | y] padded to <= 64
constexpr int WLS_THREADS = 256;

// partial[wg][dp*dp]: Gram matrix of this workgroup's rows (dp = padded d+1, multiple of 16)
__global__ __launch_bounds__(WLS_THREADS) void wls_gram_kernel(const double *__restrict__ X,
                                                               const double *__restrict__ y,
                                                               const double *__restrict__ w,
                                                               int64_t n, int d, int dp,
                                                               double *__restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = dp / 16;                                    // 16x16 blocks per side (<= 4)
    const int i = lane & 15, kk = lane >> 4;
    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // a k-step covers 4 consecutive rows; waves of all workgroups stride over the k-steps
    const int64_t steps = (n + 3) / 4;
    const int64_t wid = (int64_t)blockIdx.x * (WLS_THREADS / 64) + wave;
    const int64_t nwaves = (int64_t)gridDim.x * (WLS_THREADS / 64);
    for (int64_t s = wid; s < steps; s += nwaves) {
        const int64_t row = s * 4 + kk;
        const bool ok = row < n;
        const double wr = ok ? w[row] : 0.0;
        double xa[4], xb[4];                                   // Xa[row][blk*16 + i], weighted / plain
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int col = blk * 16 + i;
            double v = 0.0;
            if (ok && blk < nb) v = col < d ? X[row * d + col] : (col == d ? y[row] : 0.0);
            xb[blk] = v;
            xa[blk] = wr * v;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (a < nb && b < nb && b >= a)                // upper block triangle (symmetric)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a], xb[b], acc[a][b], 0, 0, 0);
    }
    // cross-wave sum through LDS, then one partial matrix per workgroup
    __shared__ double sh[64 * 64];
    for (int e = threadIdx.x; e < dp * dp; e += WLS_THREADS) sh[e] = 0.0;
    __syncthreads();
    for (int wv = 0; wv < WLS_THREADS / 64; ++wv) {            // fixed wave order: deterministic
        if (wave == wv) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (a < nb && b < nb && b >= a) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rr = a * 16 + (lane >> 4) + 4 * r, cc = b * 16 + (lane & 15);
                            sh[rr * dp + cc] += acc[a][b][r];
                        }
                    }
        }
        __syncthreads();
    }
    double *out = partial + (size_t)blockIdx.x * dp * dp;
    for (int e = threadIdx.x; e < dp * dp; e += WLS_THREADS) out[e] = sh[e];
}

// One workgroup: sum the partials (fixed order), Cholesky of G = X^T W X, solve G theta = X^T W y.
__global__ __launch_bounds__(64) void wls_solve_kernel(const double *__restrict__ partial, int nparts,
                                                       int d, int dp, double *__restrict__ theta,
                                                       int32_t *__restrict__ status) {
    __shared__ double G[64][65];
    __shared__ double rhs[64];
    const int t = threadIdx.x;
    for (int e = t; e < dp * dp; e += 64) {
        double s = 0.0;
        for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * dp * dp + e];
        G[e / dp][e % dp] = s;
    }
    __syncthreads();
    if (t < d) rhs[t] = G[t][d];                               // X^T W y is column d of the upper part
    __syncthreads();
    // upper triangle was accumulated; mirror it
    for (int e = t; e < d * d; e += 64) {
        const int r = e / d, c = e % d;
        if (r > c) G[r][c] = G[c][r];
    }
    __syncthreads();
    // right-looking Cholesky, column j handled by all threads (d <= 63: rows t > j)
    bool bad = false;
    for (int j = 0; j < d; ++j) {
        const double piv = G[j][j];
        if (!(piv > 0.0)) { bad = true; break; }               // uniform: every thread reads G[j][j]
        const double dj = sqrt(piv);
        __syncthreads();
        if (t == j) G[j][j] = dj;
        if (t > j && t < d) G[t][j] = G[t][j] / dj;
        __syncthreads();
        if (t > j && t < d) {
            const double ltj = G[t][j];
            for (int c = j + 1; c <= t; ++c) G[t][c] -= ltj * G[c][j];
        }
        __syncthreads();
    }
    if (bad) {
        if (t == 0) atomicOr(status, RLVI_ST_SINGULAR);
        if (t < d) theta[t] = __builtin_nan("");
        return;
    }
    // L z = rhs, then L^T theta = z (serial in j, parallel updates)
    for (int j = 0; j < d; ++j) {
        if (t == j) rhs[j] = rhs[j] / G[j][j];
        __syncthreads();
        if (t > j && t < d) rhs[t] -= G[t][j] * rhs[j];
        __syncthreads();
    }
    for (int j = d - 1; j >= 0; --j) {
        if (t == j) rhs[j] = rhs[j] / G[j][j];
        __syncthreads();
        if (t < j) rhs[t] -= G[j][t] * rhs[j];
        __syncthreads();
    }
    if (t < d) theta[t] = rhs[t];
}

}  // namespace rlvi

using namespace rlvi;

extern "C" int rlvi_wls_solve_f64(const double *X, const double *y, const double *w, int64_t n,
                                  int64_t d, double *theta, void *ws, void *stream) {
    if (!X || !y || !w || !theta || !ws) return RLVI_E_NULL;
    if (n <= 0 || d <= 0) return RLVI_E_SHAPE;
    if (d > WLS_MAXD) return RLVI_E_LIMIT;
    if (((uintptr_t)X & 7) || ((uintptr_t)y & 7) || ((uintptr_t)w & 7) || ((uintptr_t)theta & 7) ||
        ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int dp = (int)(((d + 1) + 15) / 16) * 16;
    int nwg = (int)((n + 1023) / 1024);
    if (nwg > WLS_MAX_WG) nwg = WLS_MAX_WG;
    double *partial = reinterpret_cast<double *>(static_cast<char *>(ws) + WS_WLS_OFF);
    int32_t *status = reinterpret_cast<int32_t *>(ws);
    const int rc = launch(wls_gram_kernel, dim3(nwg), dim3(WLS_THREADS), 0, st, X, y, w, n, (int)d, dp,
                          partial);
    if (rc != 0) return rc;
    return launch(wls_solve_kernel, dim3(1), dim3(64), 0, st, partial, nwg, (int)d, dp, theta, status);
}
