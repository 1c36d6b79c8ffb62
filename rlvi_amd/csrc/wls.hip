// Weighted least squares on the device: the M-step of the linear-regression path.
//
// Replaces  theta = lstsq(diag(sqrt(w)) @ X, diag(sqrt(w)) @ y)  (standard-learning/rlvi.py:70-71,
// :79-80; the reference materialises the n x n diagonal and calls scipy's LAPACK gelsd) by the
// normal equations  (Xa^T W Xa) with Xa = [X | y]:  the (d+1) x (d+1) weighted Gram matrix is THE
// dense contraction of this path (n*d^2 flops) and runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64); a Cholesky factorisation + two triangular solves of the d x d system
// finish in one workgroup.  A rank-deficient design (a non-positive or negligible Cholesky pivot)
// sets RLVI_ST_SINGULAR and hands over to wls_minnorm_kernel: the reference's lstsq (LAPACK gelsd)
// returns the MINIMUM-NORM least-squares solution there, which is pinv(G) X^T W y -- a cyclic Jacobi
// eigen-decomposition of the d x d Gram matrix in one wave, eigenvalues below d * 64 eps * lambda_max
// dropped (G carries the design's singular values squared, so exact rank deficiency -- duplicated or
// empty columns, too few weighted rows -- is what this resolves; sigma_min / sigma_max below ~1e-7 is
// treated as rank-deficient, where gelsd on the design itself would still resolve it).
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
                                                       int32_t *__restrict__ status,
                                                       int32_t *__restrict__ minnorm_flag) {
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
    double dmax = 0.0;
    for (int j = 0; j < d; ++j) dmax = G[j][j] > dmax ? G[j][j] : dmax;
    const double piv_min = dmax * (double)d * 64.0 * 2.220446049250313e-16;
    for (int j = 0; j < d; ++j) {
        const double piv = G[j][j];
        if (!(piv > piv_min)) { bad = true; break; }           // uniform: every thread reads G[j][j]
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
        // rank-deficient (or not finite): wls_minnorm_kernel, enqueued behind this one, takes over
        if (t == 0) { atomicOr(status, RLVI_ST_SINGULAR); *minnorm_flag = 1; }
        if (t < d) theta[t] = __builtin_nan("");
        return;
    }
    if (t == 0) *minnorm_flag = 0;
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

// Minimum-norm solution theta = pinv(G) b for a rank-deficient Gram matrix (runs only when the
// Cholesky kernel raised the flag).  One wave: cyclic Jacobi sweeps G <- J^T G J, V <- V J until the
// off-diagonal mass is negligible, then theta = sum over the kept eigenpairs of v (v.b) / lambda.
__global__ __launch_bounds__(64) void wls_minnorm_kernel(const double *__restrict__ partial, int nparts,
                                                         int d, int dp, double *__restrict__ theta,
                                                         const int32_t *__restrict__ minnorm_flag) {
    if (*minnorm_flag == 0) return;
    extern __shared__ double sm[];
    double *G = sm;                 // [64][64]
    double *V = sm + 64 * 64;       // [64][64]
    double *b = sm + 2 * 64 * 64;   // [64]
    double *red = b + 64;           // [64]
    const int t = threadIdx.x;
    for (int e = t; e < 64 * 64; e += 64) { G[e] = 0.0; V[e] = (e / 64 == e % 64) ? 1.0 : 0.0; }
    __syncthreads();
    for (int e = t; e < dp * dp; e += 64) {
        const int r = e / dp, c = e % dp;
        if (r <= c && c <= d && r < d) {
            double s = 0.0;
            for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * dp * dp + e];
            if (c == d) b[r] = s;
            else { G[r * 64 + c] = s; G[c * 64 + r] = s; }
        }
    }
    __syncthreads();
    bool finite = true;
    for (int c = 0; c < d; ++c) finite = finite && (t >= d || (G[t * 64 + c] - G[t * 64 + c] == 0.0));
    finite = finite && (t >= d || (b[t] - b[t] == 0.0));
    if (!__all(finite)) {                                       // NaN / inf in the data: NaN, as lstsq raises
        if (t < d) theta[t] = __builtin_nan("");
        return;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
        if (t < d) {
            for (int c = 0; c < d; ++c) { const double v = G[t * 64 + c]; if (c == t) dg += v * v; else off += v * v; }
        }
        off = wave_sum(off);
        dg = wave_sum(dg);
        if (off <= 1e-60 * dg || off == 0.0) break;
        for (int p = 0; p < d - 1; ++p) {
            for (int q = p + 1; q < d; ++q) {
                const double apq = G[p * 64 + q];
                if (apq == 0.0) continue;                        // uniform: every lane reads the same word
                const double app = G[p * 64 + p], aqq = G[q * 64 + q];
                const double tau = (aqq - app) / (2.0 * apq);
                const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + tt * tt), sn = tt * c;
                __syncthreads();
                if (t < d) {                                     // columns p, q of G and V
                    const double gp = G[t * 64 + p], gq = G[t * 64 + q];
                    G[t * 64 + p] = c * gp - sn * gq;
                    G[t * 64 + q] = sn * gp + c * gq;
                    const double vp = V[t * 64 + p], vq = V[t * 64 + q];
                    V[t * 64 + p] = c * vp - sn * vq;
                    V[t * 64 + q] = sn * vp + c * vq;
                }
                __syncthreads();
                if (t < d) {                                     // rows p, q of G
                    const double gp = G[p * 64 + t], gq = G[q * 64 + t];
                    G[p * 64 + t] = c * gp - sn * gq;
                    G[q * 64 + t] = sn * gp + c * gq;
                }
                __syncthreads();
                if (t == 0) { G[p * 64 + q] = 0.0; G[q * 64 + p] = 0.0; }
                __syncthreads();
            }
        }
    }
    // lane t = eigenpair t
    const double lam = t < d ? G[t * 64 + t] : 0.0;
    const double lmax = wave_max(lam);
    const double cut = lmax * (double)d * 64.0 * 2.220446049250313e-16;
    double coef = 0.0;
    if (t < d && lam > cut) {
        double vb = 0.0;
        for (int r = 0; r < d; ++r) vb += V[r * 64 + t] * b[r];
        coef = vb / lam;
    }
    red[t] = coef;
    __syncthreads();
    if (t < d) {
        double th = 0.0;
        for (int i = 0; i < d; ++i) th += V[t * 64 + i] * red[i];
        theta[t] = th;
    }
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
    // (the min-norm kernel is a no-op unless the Cholesky kernel raised the flag)
    int32_t *flag = &static_cast<WsHeader *>(ws)->wls_minnorm;
    int rc2 = launch(wls_solve_kernel, dim3(1), dim3(64), 0, st, partial, nwg, (int)d, dp, theta, status, flag);
    if (rc2 != 0) return rc2;
    constexpr size_t MINNORM_LDS = (2 * 64 * 64 + 128) * sizeof(double);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(wls_minnorm_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)MINNORM_LDS);
    if (attr != hipSuccess) return (int)attr;
    return launch(wls_minnorm_kernel, dim3(1), dim3(64), MINNORM_LDS, st, partial, nwg, (int)d, dp, theta,
                  (const int32_t *)flag);
}
