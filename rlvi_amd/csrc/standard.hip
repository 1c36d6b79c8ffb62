// The standard-learning / online-learning estimators as ONE launch each.
//
//   rlvi_linear_regression_f64      standard-learning/rlvi.py:68-89   linear_regression(X, y, maxiter, tol)
//   rlvi_sample_weight_online_f64   online-learning/main.py:293-297   log_proba -> residuals -> update_weights_rlvi
//
// The reference alternates a closed-form E-step (update_weights, rlvi.py:8-20: up to 100 population-wide
// reductions) with a weighted least-squares fit (scipy lstsq on diag(sqrt(w))-scaled rows, rlvi.py:79-80) until
// theta stops moving (rlvi.py:85-87) -- at n = 1000, d = 20 three outer iterations, 135 inner ones and four
// solves of a 20 x 20 system: 40 kflop per product, nothing a launch per step or a host decision per outer
// iteration could ever amortise (round 3: six launches + three torch ops + one host sync per outer iteration,
// 0.62 ms per call).  Here the whole estimator is one persistent workgroup of four waves (one per SIMD):
//   * the samples live in registers (n <= 4096: up to sixteen per thread) and in two LDS vectors (weights, squared
//     residuals); every reduction of the fixed point is a wave butterfly + ONE workgroup barrier (partials in
//     parity-buffered LDS slots, summed in wave order by everybody: all threads take the same stop decision);
//   * the weighted Gram matrix [X | y]^T W [X | y] -- THE dense contraction of the path -- runs on the fp64 matrix
//     cores (v_mfma_f64_16x16x4_f64), sixteen 4-row k-panels of [X | y] per wave and trip with all their loads in
//     flight together (the design stays in the L2); X.theta likewise (16 rows per wave and instruction, k over
//     the columns);
//   * the (d+1) x (d+1) system is factored (L D L^T) and solved by wave 0 with a row per lane in registers and
//     v_readlane broadcasts -- no LDS round trip and no barrier per pivot;
//   * the stop test ||theta - prev|| / ||prev|| <= tol is evaluated by every wave from LDS: no host in the loop,
//     ONE launch, and the host waits once for theta.
// A pivot that says "rank-deficient" (the reference's lstsq then returns the minimum-norm solution) ends the
// launch with info[3] = 1: the caller runs the general path (wls.hip: Jacobi pseudo-inverse) instead.
// Sums are fp64 in a fixed order: same inputs, same bits.
#include <string.h>

#include "rlvi_common.h"

namespace rlvi {

typedef double sd4_t __attribute__((ext_vector_type(4)));

// (four waves, one per SIMD: a reduction of the fixed point costs every WAVE ~80 instructions whatever it holds, and
//  two waves on a SIMD take turns at its issue slots -- eight waves with two samples per thread measured 0.79 us per
//  inner iteration)
constexpr int SL_THREADS = 256;
constexpr int SL_NW = SL_THREADS / WAVE;
constexpr int SL_DP = 32;                         // [X | y] padded to two 16-column blocks: d <= 31
constexpr int SL_XS = 64;                          // 4-row k-panels a wave keeps in registers (resident design: n <= 1024)
constexpr int SL_RU = 4;                           // 16-row blocks of the residual pass in flight per wave and trip
constexpr int SL_GU = 16;                          // 4-row k-panels of the Gram loop in flight per wave and trip
constexpr int SL_MAXN = 4096;                     // samples the one-workgroup form takes (16 per thread)
constexpr int SL_GPITCH = SL_DP + 1;

// 1 / x to working precision: v_rcp_f64 + two Newton steps (the factorisation's pivots; not correctly rounded,
// 1e-16 relative, and a tenth of the IEEE division's instruction count on a wave that issues alone)
__device__ __forceinline__ double sl_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// the value is needed HERE, whatever later selects do with it: keeps the compiler from sinking a load into the
// branch of the select that consumes it (a load under a branch is waited for before the next one is issued)
__device__ __forceinline__ void sl_keep(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ double sl_readlane(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xFFFFFFFFll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// {a, b} summed over the workgroup, every thread gets the totals: wave butterflies, one slot pair per wave in
// `slots` (two parities x SL_NW x 2 doubles), ONE barrier; the slots of parity p are not written again before
// everybody has passed the NEXT barrier, i.e. has read them.
__device__ __forceinline__ void sl_block_sum2(double &a, double &b, double *slots, int &parity) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    a = wave_sum(a);
    b = wave_sum(b);
    double *s = slots + (size_t)parity * SL_NW * 2;
    if (lane == 0) { s[2 * wave] = a; s[2 * wave + 1] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int w = 0; w < SL_NW; ++w) { ta += s[2 * w]; tb += s[2 * w + 1]; }
    a = ta; b = tb;
    parity ^= 1;
}

// lab build (-DRLVI_STAMPS=1, tools/build_variants.py): thread 0 leaves 100 MHz wall-clock stamps of the phases in
// the workspace scratch (tools/lab/linreg_phases.py prints them); compiled out of the product library
#if RLVI_STAMPS
#define SL_STAMP() do { if (threadIdx.x == 0 && sh.nst < 120) sh.dbg[sh.nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SL_STAMP() do { } while (0)
#endif

struct SlShared {
#if RLVI_STAMPS
    mutable unsigned long long *dbg;
    mutable int nst;
#endif
    double *wsh;        // [npad]  weights (pads 0)
    double *rsh;        // [npad]  squared residuals
    double *red;        // [SL_NW][3][256]  Gram partials of the waves
    double *G;          // [SL_DP][SL_GPITCH]
    double *LT;         // [SL_DP][SL_GPITCH]  L^T of the factorisation (row j = column j of L)
    double *th;         // [SL_DP] theta, [SL_DP] previous theta
    double *rhs;        // [SL_DP] X^T W y
    double *slots;      // [2][SL_NW][2]
    int *flag;          // [4]
    double *x1s;        // [1024][8]  columns 16 .. 23 of [X | y] (resident design with two column blocks)
};

// theta = argmin sum_i w_i (y_i - x_i.theta)^2 from the weights in sh.wsh -> sh.th[0 .. d).  Returns false (on
// every thread) when a pivot is not safely positive.  All threads call.  DS: the system is solved padded to DS >= d
// rows -- straight-line code, no guard inside the factorisation; NB: 16-column blocks of [X | y] (2 when d >= 16).
// RES: the design is RESIDENT on the chip (n <= 1024, d <= 23): lane (i, kk) of wave w keeps
// [X | y][4 (w + 4 k) + kk][i], k < 64, in registers (xr) and columns 16 .. 23 live in LDS (sh.x1s) -- no pass over
// global memory after the first.
template <int DS, int NB, bool RES>
__device__ __forceinline__ bool sl_wls(const SlShared &sh, const double *__restrict__ X, const double *__restrict__ y,
                                       int n, int d, const double (&xr)[RES ? SL_XS : 1]) {
    constexpr int DP = NB * 16;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int i = lane & 15, kk = lane >> 4;
    sd4_t a00 = {0.0, 0.0, 0.0, 0.0}, a01 = a00, a11 = a00;
    const int steps = (n + 3) / 4;
    if constexpr (RES) {
        const int c1 = (i & 7) * 1;
#pragma unroll
        for (int k0 = 0; k0 < SL_XS; k0 += 8) {
            if (4 * (wave + SL_NW * k0) < n) {                        // (wave-uniform: a short design skips its empty blocks)
                double w[8], x1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = 4 * (wave + SL_NW * (k0 + u)) + kk;      // < 1024 = the padded length of wsh / x1s
                    w[u] = sh.wsh[row];                              // rows past n: weight 0, panel 0
                    x1[u] = NB > 1 ? sh.x1s[row * 8 + c1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double x0 = xr[k0 + u];
                    const double w0 = w[u] * x0;
                    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, x0, a00, 0, 0, 0);
                    if (NB > 1) {
                        const double xb = i < 8 ? x1[u] : 0.0;
                        a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, xb, a01, 0, 0, 0);
                        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u] * xb, xb, a11, 0, 0, 0);
                    }
                }
            }
        }
    } else
    // SL_GU k-panels (4 rows each) per trip, all their loads in flight together: the design comes from the L2
    // (160 KB at n = 1000, d = 20) and a trip's registers are free again before the factorisation needs its own.
    // One CU streams from the L2 at ~45 GB/s (its outstanding misses x the L2's latency), so a pass over the
    // design is microseconds whatever the loop looks like: in-kernel stamps (tools/lab/linreg_phases.py) put this
    // pass and the residual pass at the top of the launch's time beside the E-step's serial reductions.
    for (int s0 = wave; s0 < steps; s0 += SL_GU * SL_NW) {      // (the body of `else` when RES)
        double x0[SL_GU], x1[SL_GU], w[SL_GU];
        int xrow0[SL_GU];
#pragma unroll
        for (int u = 0; u < SL_GU; ++u) {
            const int s = s0 + u * SL_NW;
            const int row = 4 * s + kk;
            const bool ok = s < steps && row < n;
            const int rr = ok ? row : 0;
            // every lane that HAS a column (its column of X, or y) loads from an always valid row, and the value
            // is selected afterwards; the lanes without one (5 of a second block's 16 carry data at d = 20) ask for
            // nothing.  One branch around ALL loads of the trip: a load under a branch of its own is waited for
            // before the next one is issued (the first version of this loop: 64 dependent round trips per trip)
            xrow0[u] = rr;
            w[u] = sh.wsh[rr];
            x0[u] = 0.0;
            x1[u] = 0.0;
        }
        if (i <= d) {
#pragma unroll
            for (int u = 0; u < SL_GU; ++u) x0[u] = *(i < d ? X + (size_t)xrow0[u] * d + i : y + xrow0[u]);
        }
        if (DP > 16 && 16 + i <= d) {
#pragma unroll
            for (int u = 0; u < SL_GU; ++u) x1[u] = *(16 + i < d ? X + (size_t)xrow0[u] * d + 16 + i : y + xrow0[u]);
        }
        __builtin_amdgcn_sched_barrier(0);        // every load of the trip is out before the first value is used
#pragma unroll
        for (int u = 0; u < SL_GU; ++u) {
            // (the predicates again, from the indices: kept across the barrier they are 48 lane masks)
            const int s = s0 + u * SL_NW;
            const bool ok = s < steps && 4 * s + kk < n;
            sl_keep(x0[u]);
            if (DP > 16) sl_keep(x1[u]);
            x0[u] = (ok && i <= d) ? x0[u] : 0.0;
            x1[u] = (DP > 16 && ok && 16 + i <= d) ? x1[u] : 0.0;
            w[u] = ok ? w[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < SL_GU; ++u) {
            const double w0 = w[u] * x0[u], w1 = w[u] * x1[u];
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, x0[u], a00, 0, 0, 0);
            if (DP > 16) {
                a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, x1[u], a01, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, x1[u], a11, 0, 0, 0);
            }
        }
    }
    SL_STAMP();   // gram loop done
    // the waves' partial blocks -> LDS; element e of a block = (result r, lane): row (lane >> 4) + 4 r, column lane & 15
    constexpr int NBLK = NB > 1 ? 3 : 1;
    double *my = sh.red + (size_t)wave * 3 * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        my[0 * 256 + r * 64 + lane] = a00[r];
        if (DP > 16) {
            my[1 * 256 + r * 64 + lane] = a01[r];
            my[2 * 256 + r * 64 + lane] = a11[r];
        }
    }
    __syncthreads();
    // block sums in wave order (deterministic) -> the system PADDED WITH THE IDENTITY to DP rows (a padded row has
    // pivot 1 and no coupling: the factorisation below is straight-line code without a guard), the right-hand
    // side X^T W y (column d) set aside
    for (int e = tid; e < NBLK * 256; e += SL_THREADS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SL_NW; ++w) s += sh.red[(size_t)w * 3 * 256 + e];
        const int blk = e >> 8, r = (e >> 6) & 3, l = e & 63;
        const int row = (blk == 2 ? 16 : 0) + (l >> 4) + 4 * r, col = (blk >= 1 ? 16 : 0) + (l & 15);
        const double v = (row < d && col < d) ? s : (row == col ? 1.0 : 0.0);
        sh.G[row * SL_GPITCH + col] = v;
        if (blk == 1) sh.G[col * SL_GPITCH + row] = v;
        if (col == d && row < d) sh.rhs[row] = s;      // (ONE writer per entry: the mirrored element of a diagonal
                                                       //  block is another rounding of the same sum)
    }
    __syncthreads();
    SL_STAMP();   // block sums done
    // ---- wave 0: G = L D L^T, row t of the lower triangle in lane t's registers
    if (wave == 0) {
        int t = lane & (SL_DP - 1);
        // (opaque: the lane masks t > j, t == j below are loop-invariant across the estimator's outer loop, and
        //  hoisted there they are 2 DP scalar register pairs that spill)
        asm volatile("" : "+v"(t));
        double g[DS];
#pragma unroll
        for (int c = 0; c < DS; ++c) g[c] = sh.G[t * SL_GPITCH + c];
        double b = t < d ? sh.rhs[t < SL_DP ? t : 0] : 0.0;
        double dmax = t < d ? sh.G[t * SL_GPITCH + t] : 0.0;
        dmax = wave_max(dmax);
        const double piv_min = dmax * (double)d * 64.0 * 2.220446049250313e-16;
        bool bad = !(dmax == dmax);
        double dinv = 0.0;                       // lane j: 1 / D_j
#pragma unroll
        for (int j = 0; j < DS; ++j) {
            const double piv = sl_readlane(g[j], j);
            bad = bad || (j < d && !(piv > piv_min));
            const double inv = sl_rcp(piv);
            const double l = g[j] * inv;                              // L[t][j] for t > j
            dinv = t == j ? inv : dinv;
#pragma unroll
            for (int c = j + 1; c < DS; ++c) {
                const double lc = sl_readlane(g[j], c);              // G[c][j] = L[c][j] D_j
                g[c] = __builtin_fma(-l, lc, g[c]);                  // (lanes t < c compute values nobody reads)
                // (eight broadcasts at a time: all 31 of a pivot hoisted at once do not fit the scalar registers)
                if (((c - j) & 7) == 0) __builtin_amdgcn_sched_barrier(0);
            }
            g[j] = t > j ? l : 0.0;                                   // strictly lower part of L; 0 elsewhere
            sh.LT[j * SL_GPITCH + t] = g[j];                          // row j of L^T: for the back substitution
            __builtin_amdgcn_sched_barrier(0);
        }
        // L z = b (lane t ends with z_t; g[j] = 0 for t <= j: no mask), then z / D
#pragma unroll
        for (int j = 0; j < DS; ++j) b = __builtin_fma(-g[j], sl_readlane(b, j), b);
        b *= dinv;
        // L^T theta = z: lane t needs column t of L = row t of L^T (0 for c <= t: no mask)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < DS; ++c) g[c] = sh.LT[t * SL_GPITCH + c];
#pragma unroll
        for (int j = DS - 1; j >= 1; --j) b = __builtin_fma(-g[j], sl_readlane(b, j), b);
        if (lane < d) sh.th[lane] = b;
        if (lane == 0) sh.flag[0] = bad ? 1 : 0;
    }
    __syncthreads();
    SL_STAMP();   // solved
    return sh.flag[0] == 0;
}

// r_i = (y_i - x_i.theta)^2 -> sh.rsh, sigma2 = w.r / sum(w); returns sigma2 on every thread
template <int NB, bool RES>
__device__ __forceinline__ double sl_residuals(const SlShared &sh, const double *__restrict__ X,
                                               const double *__restrict__ y, int n, int d, int &parity,
                                               const double (&xr)[RES ? SL_XS : 1]) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int i = lane & 15, kk = lane >> 4;
    double num = 0.0, den = 0.0;
    if constexpr (RES) {
        // from the resident panels: lane (i, kk) holds [X | y][row][i] (and reads [row][16 + i], i < 8, from LDS); with
        // c = [theta; -1 in column d] the row's x.theta - y is the sum over the 16 lanes of a row's group -- four
        // DPP steps -- and its square is the residual.  No matrix instruction (they contract over the ROWS of a
        // panel, which is the Gram matrix's sum), no byte from global memory.
        const double c0 = i < d ? sh.th[i] : (i == d ? -1.0 : 0.0);
        const int j1 = 16 + (i & 7);
        const double c1 = (NB > 1 && i < 8) ? (j1 < d ? sh.th[j1 < SL_DP ? j1 : 0] : (j1 == d ? -1.0 : 0.0)) : 0.0;
#pragma unroll
        for (int k0 = 0; k0 < SL_XS; k0 += 8) {
            // eight panels per block of straight-line code (their LDS reads and butterflies interleave); a short
            // design skips its empty blocks (wave-uniform)
            if (4 * (wave + SL_NW * k0) < n) {
                double p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = 4 * (wave + SL_NW * (k0 + u)) + kk;
                    p[u] = xr[k0 + u] * c0;
                    if (NB > 1) p[u] = __builtin_fma(sh.x1s[row * 8 + (i & 7)], c1, p[u]);
                }
                // the eight partials of a lane -> the eight row sums of its 16-lane group, TRANSPOSING on the way:
                // after the exchange with lane ^ 1 a lane keeps the panels of its own parity (four values), after
                // lane ^ 2 two, and those two are finished by two rotations of the row -- 54 cross-lane instructions for
                // eight panels instead of 96; lane (b2 b1) of a group ends with the sums of panels b2b1 and 4 + b2b1
                const bool b1 = (i & 1) != 0, b2 = (i & 2) != 0;
                double q[4], r[2];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double keep = b1 ? p[2 * m + 1] : p[2 * m], send = b1 ? p[2 * m] : p[2 * m + 1];
                    q[m] = keep + dpp_x<0xB1>(send);                 // quad_perm [1,0,3,2]
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const double keep = b2 ? q[2 * m + 1] : q[2 * m], send = b2 ? q[2 * m] : q[2 * m + 1];
                    r[m] = keep + dpp_x<0x4E>(send);                 // quad_perm [2,3,0,1]
                    // (the partners of the last two steps must share the lane's low two bits: rotations by 8 and by 4
                    //  inside the 16-lane row, not the mirrors)
                    r[m] = r[m] + dpp_x<0x128>(r[m]);                // row_ror:8
                    r[m] = r[m] + dpp_x<0x124>(r[m]);                // row_ror:4
                }
                // lanes 0 .. 3 of a group keep the sums and write the residuals of panels i and 4 + i
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int u = 4 * m + (i & 3);
                    const int row = 4 * (wave + SL_NW * (k0 + u)) + kk;
                    const double rr = r[m] * r[m];
                    const double wr = i < 4 ? sh.wsh[row] : 0.0;
                    if (i < 4) sh.rsh[row] = rr;
                    num = __builtin_fma(wr, rr, num);
                    den += wr;
                }
            }
        }
        sl_block_sum2(num, den, sh.slots, parity);
        SL_STAMP();   // residuals done
        return num / den;
    }
    const int blocks = (n + 15) / 16;
    // X[16 rb .. +15] . theta on the matrix cores: A = a 16 x 4 panel of X, B = the 4 matching entries of theta in
    // every column; lane l's results are rows (l >> 4) + 4 r, all columns alike.  SL_RU row blocks per wave and
    // trip, all their loads in flight together (one block per trip was one L2 round trip per 16 rows: 11 us per
    // pass at n = 1000).
    double bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                    // d <= 31: eight 4-column panels at most
        const int k = 4 * u + kk;
        const double tv = sh.th[k < d ? k : 0];
        bv[u] = k < d ? tv : 0.0;
    }
    for (int rb0 = wave; rb0 < blocks; rb0 += SL_RU * SL_NW) {
        double a[SL_RU][8], yv[SL_RU];
#pragma unroll
        for (int q = 0; q < SL_RU; ++q) {
            const int rb = rb0 + q * SL_NW;
            const int row = 16 * rb + i;
            const double *xrow = X + (size_t)(row < n ? row : 0) * d;
#pragma unroll
            for (int u = 0; u < 8; ++u) a[q][u] = xrow[(4 * u + kk) < d ? 4 * u + kk : 0];      // (branch-free)
            const int r = 16 * rb + kk + 4 * (i & 3);
            yv[q] = y[r < n ? r : 0];
        }
        __builtin_amdgcn_sched_barrier(0);        // every load of the trip is out before the first value is used
#pragma unroll
        for (int q = 0; q < SL_RU; ++q) {
            const int rb = rb0 + q * SL_NW;
            const bool rok = 16 * rb + i < n;
            sd4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sl_keep(a[q][u]);
                const double av = (rok && 4 * u + kk < d) ? a[q][u] : 0.0;
                if (4 * u < d) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u], acc, 0, 0, 0);
            }
            sl_keep(yv[q]);
            const int r = 16 * rb + kk + 4 * i;
            if (i < 4 && r < n) {
                const double p = i == 0 ? acc[0] : i == 1 ? acc[1] : i == 2 ? acc[2] : acc[3];
                const double e = yv[q] - p, rr = e * e;
                const double w = sh.wsh[r];
                sh.rsh[r] = rr;
                num = __builtin_fma(w, rr, num);
                den += w;
            }
        }
    }
    sl_block_sum2(num, den, sh.slots, parity);      // (its barrier also publishes rsh)
    SL_STAMP();   // residuals done
    return num / den;
}

template <int E, int DS, int NB, bool RES>
__global__ __launch_bounds__(SL_THREADS) void linreg_rlvi_kernel(
    const double *__restrict__ X, const double *__restrict__ y, int n, int d, int npad, int maxiter, double tol,
    double etol, int emaxiter, double *__restrict__ theta_out, double *__restrict__ w_out,
    int32_t *__restrict__ info, unsigned long long *__restrict__ dbg) {
    extern __shared__ double sm[];
    SlShared sh;
#if RLVI_STAMPS
    sh.dbg = dbg;
    sh.nst = 0;
#else
    (void)dbg;
#endif
    sh.wsh = sm;
    sh.rsh = sh.wsh + npad;
    sh.red = sh.rsh + npad;
    sh.G = sh.red + SL_NW * 3 * 256;
    sh.LT = sh.G + SL_DP * SL_GPITCH;
    sh.th = sh.LT + SL_DP * SL_GPITCH;
    sh.rhs = sh.th + 2 * SL_DP;
    sh.slots = sh.rhs + SL_DP;
    sh.flag = reinterpret_cast<int *>(sh.slots + 2 * SL_NW * 2);
    sh.x1s = reinterpret_cast<double *>(sh.flag + 4);
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    int parity = 0;

    // resident design: the k-panels of column block 0 into registers, columns 16 .. 23 into LDS -- once
    double xr[RES ? SL_XS : 1];
    if constexpr (RES) {
        const int i = lane & 15, kk = lane >> 4;
        if (i <= d) {                                  // (one branch around all loads: they are in flight together)
#pragma unroll
            for (int k = 0; k < SL_XS; ++k) {
                const int row = 4 * (wave + SL_NW * k) + kk;
                const int rr = row < n ? row : 0;
                xr[k] = *(i < d ? X + (size_t)rr * d + i : y + rr);
            }
        } else {
#pragma unroll
            for (int k = 0; k < SL_XS; ++k) xr[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < SL_XS; ++k) xr[k] = (4 * (wave + SL_NW * k) + kk < n) ? xr[k] : 0.0;
        if (NB > 1) {
            for (int q0 = tid; q0 < 1024 * 8; q0 += 8 * SL_THREADS) {        // eight loads in flight per thread
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u * SL_THREADS;
                    const int row = q >> 3, col = 16 + (q & 7);
                    const int rr = row < n ? row : 0;
                    v[u] = *(col < d ? X + (size_t)rr * d + col : y + rr);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u * SL_THREADS;
                    const int row = q >> 3, col = 16 + (q & 7);
                    sl_keep(v[u]);
                    sh.x1s[q] = (row < n && col <= d) ? v[u] : 0.0;
                }
            }
        }
    } else {
        xr[0] = 0.0;
    }
    for (int q = tid; q < npad; q += SL_THREADS) sh.wsh[q] = q < n ? 1.0 : 0.0;     // weights = ones (rlvi.py:69)
    for (int q = tid; q < 2 * SL_DP; q += SL_THREADS) sh.th[q] = 0.0;
    for (int q = tid; q < SL_DP * SL_GPITCH; q += SL_THREADS) sh.G[q] = 0.0;
    __syncthreads();

    int outer = 0, inner_last = 0, inner_all = 0;
    bool ok = true;
    const double invn = 1.0 / (double)n;
    for (;;) {
        SL_STAMP();   // outer iteration starts
        // theta from the current weights (rlvi.py:70-71 the first time, :79-80 afterwards), then the residuals
        // and sigma2 (:72-73, :81-82)
        ok = sl_wls<DS, NB, RES>(sh, X, y, n, d, xr);
        if (!ok) break;
        const double sigma2 = sl_residuals<NB, RES>(sh, X, y, n, d, parity, xr);
        if (outer > 0) {
            // ||theta - prev|| / ||prev|| <= tol (rlvi.py:85-87): every wave from LDS, the same bits everywhere
            const double tn = lane < d ? sh.th[lane] : 0.0, tp = lane < d ? sh.th[SL_DP + lane] : 0.0;
            const double dn = wave_sum((tn - tp) * (tn - tp)), pn = wave_sum(tp * tp);
            if (sqrt(dn) / sqrt(pn) <= tol) break;
        }
        if (outer >= maxiter) break;
        ++outer;
        // ---- update_weights(losses) (rlvi.py:8-20, called at :77) on this thread's samples tid + 256 j
        double e[E], w[E];
        const double hs = 0.5 / sigma2;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int q = tid + j * SL_THREADS;
            const bool v = q < n;
            e[j] = v ? exp(-(sh.rsh[v ? q : 0] * hs)) : 0.0;      // losses = 0.5 r / sigma2 (rlvi.py:74); exp(-losses)
            w[j] = v ? 0.95 : 0.0;                                 // rlvi.py:10
        }
        double ratio;
        { const double eps = 1.0 - 0.95; ratio = eps / (1.0 - eps); }
        int it = 0;
        SL_STAMP();   // exp done
        while (it < emaxiter) {
            double sse = 0.0, sum = 0.0;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                // rlvi.py:15.  (A reciprocal + Newton + correction in place of the IEEE division, with a wave-uniform
                // branch for operands near the ends of the exponent range, and the stop test on the squares, were
                // measured: 0.70 us per iteration against 0.57 -- every ballot-and-branch is a scalar round trip on
                // a wave that issues alone.)
                double nw = e[j] / (ratio + e[j]);
                nw = (tid + j * SL_THREADS < n) ? nw : 0.0;
                const double dd = nw - w[j];
                sse = __builtin_fma(dd, dd, sse);
                sum += nw;
                w[j] = nw;
            }
            sl_block_sum2(sse, sum, sh.slots, parity);
            ++it;
            if (sqrt(sse) < etol) break;                           // rlvi.py:16-19 (assign, then break)
            const double eps = 1.0 - sum * invn;                   // rlvi.py:13-14
            ratio = eps / (1.0 - eps);
        }
        SL_STAMP();   // E-step done
        inner_last = it;
        inner_all += it;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int q = tid + j * SL_THREADS;
            if (q < n) sh.wsh[q] = w[j];
        }
        if (tid < SL_DP) sh.th[SL_DP + tid] = sh.th[tid];          // theta_prev (rlvi.py:78)
        __syncthreads();
    }
    if (ok) {
        for (int q = tid; q < n; q += SL_THREADS) w_out[q] = sh.wsh[q];
        if (tid < d) theta_out[tid] = sh.th[tid];
    }
    if (tid == 0) {
        info[0] = outer;
        info[1] = inner_last;
        info[2] = inner_all;
        info[3] = ok ? 0 : 1;          // 1: rank-deficient (or not finite) -- the caller takes the general path
    }
}

// ---------------------------------------------------------------------------------------
// One mini-batch of the online path: log_proba = log sigmoid(X w + b) (first batch: log 0.5), residuals =
// -log_proba (main.py:293-296 with :84-85), sample_weight = update_weights_rlvi(residuals) (main.py:45-58, :297).
// ---------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(SL_THREADS) void online_weight_kernel(
    const double *__restrict__ X, const double *__restrict__ wv, double b, int first, int n, int d, double tol,
    int maxiter, double *__restrict__ losses_out, double *__restrict__ out, int32_t *__restrict__ out_iters) {
    __shared__ double lsh[SL_MAXN];
    __shared__ double slots[2 * SL_NW * 2];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int i = lane & 15, kk = lane >> 4;
    int parity = 0;
    if (first) {
        for (int q = tid; q < n; q += SL_THREADS) lsh[q] = 0.6931471805599453;      // -log(0.5) (main.py:293)
    } else {
        const int blocks = (n + 15) / 16;
        for (int rb = wave; rb < blocks; rb += SL_NW) {
            const int row = 16 * rb + i;
            const bool rok = row < n;
            const double *xrow = X + (size_t)(rok ? row : 0) * d;
            sd4_t acc = {0.0, 0.0, 0.0, 0.0};
            int k0 = 0;
            for (; k0 + 32 <= d; k0 += 32) {                         // eight k-panels per trip, loads in flight together
                double a[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + kk;
                    a[u] = xrow[k];                                 // (row 0 for rows past n: valid, selected away)
                    bv[u] = wv[k];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) { sl_keep(a[u]); a[u] = rok ? a[u] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bv[u], acc, 0, 0, 0);
            }
            {
                double a[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + kk;
                    a[u] = xrow[k < d ? k : 0];                     // (branch-free loads, selected below)
                    bv[u] = wv[k < d ? k : 0];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + kk;
                    sl_keep(a[u]);
                    sl_keep(bv[u]);
                    a[u] = (rok && k < d) ? a[u] : 0.0;
                    bv[u] = k < d ? bv[u] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + 4 * u < d) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bv[u], acc, 0, 0, 0);
            }
            if (i < 4) {
                const int r = 16 * rb + kk + 4 * i;
                if (r < n) {
                    const double p = (i == 0 ? acc[0] : i == 1 ? acc[1] : i == 2 ? acc[2] : acc[3]) + b;
                    lsh[r] = p >= 0.0 ? log1p(exp(-p)) : -p + log1p(exp(p));      // -log sigmoid(p)
                }
            }
        }
    }
    __syncthreads();
    double e[E], w[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int q = tid + j * SL_THREADS;
        const bool v = q < n;
        const double l = lsh[v ? q : 0];
        if (v && losses_out != nullptr) losses_out[q] = l;
        e[j] = v ? exp(-l) : 0.0;                                   // main.py:47
        w[j] = v ? 0.5 : 0.0;                                       // main.py:48
    }
    double ratio = 0.5 / (1.0 - 0.5);
    const double invn = 1.0 / (double)n;
    int it = 0;
    while (it < maxiter) {
        double sse = 0.0, sum = 0.0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double t = ratio * e[j];                          // main.py:52 (a sample past n: e = 0, update 0)
            const double nw = t / (1.0 + t);
            const double dd = nw - w[j];
            sse = __builtin_fma(dd, dd, sse);
            sum += nw;
            w[j] = nw;
        }
        sl_block_sum2(sse, sum, slots, parity);
        ++it;
        if (sqrt(sse) < tol) break;                                 // main.py:53-56 (`new` is what is returned either way)
        const double avg = sum * invn;                              // main.py:50-51
        ratio = avg / (1.0 - avg);
    }
    // new /= max(new) * len(new) (main.py:57)
    double mx = -__builtin_inf();
#pragma unroll
    for (int j = 0; j < E; ++j)
        if (tid + j * SL_THREADS < n) mx = w[j] > mx ? w[j] : mx;
    mx = wave_max(mx);
    double *s = slots + (size_t)parity * SL_NW * 2;
    if (lane == 0) s[2 * wave] = mx;
    __syncthreads();
    mx = s[0];
#pragma unroll
    for (int q = 1; q < SL_NW; ++q) mx = s[2 * q] > mx ? s[2 * q] : mx;
    const double den = mx * (double)n;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int q = tid + j * SL_THREADS;
        if (q < n) out[q] = w[j] / den;
    }
    if (tid == 0 && out_iters != nullptr) *out_iters = it;
}

}  // namespace rlvi

using namespace rlvi;

// 0 when rlvi_linear_regression_f64 takes this shape in its one-launch form (else RLVI_E_LIMIT: the caller
// composes rlvi_wls_solve_f64 / rlvi_linreg_losses_f64 / rlvi_update_weights_f64 itself)
extern "C" int rlvi_linear_regression_check(int64_t n, int64_t d) {
    if (n <= 0 || d <= 0) return RLVI_E_SHAPE;
    return (n <= SL_MAXN && d < SL_DP) ? 0 : RLVI_E_LIMIT;
}

extern "C" int rlvi_linear_regression_f64(const double *X, const double *y, int64_t n, int64_t d, int maxiter,
                                          double tol, double estep_tol, int estep_maxiter, double *theta,
                                          double *weights, int32_t *info, void *ws, void *stream) {
    if (!X || !y || !theta || !weights || !info || !ws) return RLVI_E_NULL;
    if (n <= 0 || d <= 0 || maxiter < 0 || estep_maxiter < 0) return RLVI_E_SHAPE;
    if (n > SL_MAXN || d >= SL_DP) return RLVI_E_LIMIT;
    if (((uintptr_t)X & 7) || ((uintptr_t)y & 7) || ((uintptr_t)theta & 7) || ((uintptr_t)weights & 7) ||
        ((uintptr_t)info & 3) || ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (samples per thread: four up to n = 1024, sixteen beyond; the system solved padded to 12 / 16 / 20 / 24 / 32 rows;
    //  one or two 16-column blocks of [X | y]; the design resident on the chip -- registers + LDS -- when n <= 1024
    //  and d <= 23)
    const bool two = d >= 16;                                     // column d (= y) needs the second block
    const bool res = n <= 4 * SL_THREADS && d <= 23;
    const int npad_x = res ? 1024 : (int)((n + 63) / 64) * 64;
    const size_t lds = ((size_t)2 * npad_x + (size_t)SL_NW * 3 * 256 + 2 * SL_DP * SL_GPITCH + 3 * SL_DP +
                        2 * SL_NW * 2 + 2 + ((res && two) ? 1024 * 8 : 0)) * sizeof(double);
    auto go = [&](auto kern) {
        // (> 64 KiB of dynamic LDS: asked for once per kernel and device)
        static int attr_dev = -1;
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return (int)hipErrorInvalidDevice;
        if (attr_dev != cur) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_dev = cur;
        }
        return launch(kern, dim3(1), dim3(SL_THREADS), lds, st, X, y, (int)n, (int)d, npad_x, maxiter, tol,
                      estep_tol, estep_maxiter, theta, weights, info,
                      reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF));
    };
    if (res) {
        if (d <= 12) return go(linreg_rlvi_kernel<4, 12, 1, true>);
        if (d <= 15) return go(linreg_rlvi_kernel<4, 16, 1, true>);
        if (d <= 20) return go(linreg_rlvi_kernel<4, 20, 2, true>);
        return go(linreg_rlvi_kernel<4, 24, 2, true>);
    }
    if (n <= 4 * SL_THREADS) return go(linreg_rlvi_kernel<4, 32, 2, false>);      // d = 24 .. 31
    if (d <= 12) return go(linreg_rlvi_kernel<16, 12, 1, false>);
    if (d <= 15) return go(linreg_rlvi_kernel<16, 16, 1, false>);
    if (d <= 20) return go(linreg_rlvi_kernel<16, 20, 2, false>);
    if (d <= 24) return go(linreg_rlvi_kernel<16, 24, 2, false>);
    return go(linreg_rlvi_kernel<16, 32, 2, false>);
}

extern "C" int rlvi_sample_weight_online_f64(const double *X, const double *w, double b, int first, int64_t n,
                                             int64_t d, double tol, int maxiter, double *losses_out,
                                             double *sample_weight, int32_t *out_iters, void *stream) {
    if (!sample_weight || (!first && (!X || !w))) return RLVI_E_NULL;
    if (n <= 0 || d <= 0 || maxiter < 0) return RLVI_E_SHAPE;
    if (n > SL_MAXN) return RLVI_E_LIMIT;
    if (((uintptr_t)X & 7) || ((uintptr_t)w & 7) || ((uintptr_t)sample_weight & 7) || ((uintptr_t)losses_out & 7))
        return RLVI_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define RLVI_OW(E_)                                                                                         \
    return launch(online_weight_kernel<E_>, dim3(1), dim3(SL_THREADS), 0, st, X, w, b, first, (int)n, (int)d, \
                  tol, maxiter, losses_out, sample_weight, out_iters)
    if (n <= SL_THREADS) RLVI_OW(1);
    if (n <= 2 * SL_THREADS) RLVI_OW(2);
    if (n <= 4 * SL_THREADS) RLVI_OW(4);
    if (n <= 8 * SL_THREADS) RLVI_OW(8);
    RLVI_OW(16);
#undef RLVI_OW
}
