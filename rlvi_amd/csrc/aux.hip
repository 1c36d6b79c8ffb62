// Workspace management, error strings, the in-batch E+M composition and the fp64 X.theta
// per-sample NLL kernels of the linear / logistic paths.
#include <string.h>

#include "rlvi_common.h"

using namespace rlvi;

extern "C" int rlvi_abi_version(void) { return RLVI_ABI_VERSION; }

extern "C" const char *rlvi_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case RLVI_E_NULL: return "required pointer is NULL";
        case RLVI_E_SHAPE: return "negative or inconsistent size";
        case RLVI_E_ALIGN: return "pointer or leading dimension misaligned";
        case RLVI_E_WS: return "workspace too small or not initialised";
        case RLVI_E_LIMIT: return "size beyond what the kernels support";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown rlvi error";
    }
}

extern "C" size_t rlvi_workspace_bytes(int64_t max_n, int64_t max_b) {
    return ws_bytes_for(max_n, max_b);
}

namespace rlvi {
void peers_forget(const void *ws);      // peer.hip
__global__ void ws_header_kernel(WsHeader *hdr, unsigned long long spin_ticks, int clear_status) {
    if (clear_status) hdr->status = 0;
    else hdr->spin_ticks = spin_ticks;
}
}  // namespace rlvi

extern "C" int rlvi_workspace_init(void *ws, size_t ws_bytes, void *stream) {
    if (!ws) return RLVI_E_NULL;
    if (((uintptr_t)ws & 255)) return RLVI_E_ALIGN;
    if (ws_bytes < WS_SCRATCH_OFF) return RLVI_E_WS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // control words, exchange slots, warm-start state, M-step records (the WLS / scratch regions
    // behind them need no initial value)
    hipError_t e = hipMemsetAsync(ws, 0, WS_WLS_OFF, st);
    if (e != hipSuccess) return (int)e;
    // the tagged result records of the one-launch in-batch E+M and the peer table behind them (a
    // workspace is not set up for sharded calls until rlvi_workspace_set_peers has run on it AFTER this)
    e = hipMemsetAsync(static_cast<char *>(ws) + WS_FEREC_OFF, 0, WS_FEREC_BYTES + WS_PEER_BYTES, st);
    if (e != hipSuccess) return (int)e;
    peers_forget(ws);
    ws_options_forget(ws);
    // bound of every inter-workgroup wait (RLVI_SPIN_BOUND_MS, default 100 ms), in 100 MHz ticks
    const long long ms = tune_get("RLVI_SPIN_BOUND_MS", 100);
    const unsigned long long ticks = (unsigned long long)(ms > 0 ? ms : 100) * 100000ull;
    return launch(ws_header_kernel, dim3(1), dim3(1), 0, st, static_cast<WsHeader *>(ws), ticks, 0);
}

// Byte offset (and size) of a named region of the workspace layout -- for tools and tests that look at what the
// kernels leave there: "records" (accumulate-mode M-step records), "records_out" (records of calls with `out`),
// "warm" (warm-start state of the E-step / threshold), "scratch".  (size_t)-1 for an unknown name.
extern "C" size_t rlvi_workspace_region(const char *name, size_t *bytes) {
    size_t off = (size_t)-1, n = 0;
    if (name && strcmp(name, "records") == 0) { off = WS_PART_OFF; n = WS_PART_BYTES; }
    else if (name && strcmp(name, "records_out") == 0) { off = WS_PART2_OFF; n = WS_PART_BYTES; }
    else if (name && strcmp(name, "warm") == 0) { off = WS_TRAJ_OFF; n = WS_TRAJ_BYTES; }
    else if (name && strcmp(name, "scratch") == 0) { off = WS_SCRATCH_OFF; n = 0; }
    if (bytes) *bytes = n;
    return off;
}

// Forget what earlier calls left as guesses for the next one (the E-step's trajectory and minimum, the
// threshold's key): the next E-step / threshold on this workspace starts as on a fresh one.  Results never
// depend on the guesses -- only the number of rounds / exchanges does.
extern "C" int rlvi_workspace_reset_warm(void *ws, void *stream) {
    if (!ws) return RLVI_E_NULL;
    if (((uintptr_t)ws & 255)) return RLVI_E_ALIGN;
    return (int)hipMemsetAsync(static_cast<char *>(ws) + WS_TRAJ_OFF, 0, WS_TRAJ_BYTES,
                               static_cast<hipStream_t>(stream));
}

extern "C" int rlvi_workspace_clear_status(void *ws, void *stream) {
    if (!ws) return RLVI_E_NULL;
    return launch(ws_header_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream),
                  static_cast<WsHeader *>(ws), 0ull, 1);
}

extern "C" int rlvi_workspace_status(const void *ws, int32_t *status_host, void *stream) {
    if (!ws || !status_host) return RLVI_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(status_host, ws, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(st);
}

// ---------------------------------------------------------------------------------------
// In-batch E+M (online order): NLL pass -> E-step on this batch -> weighted loss + gradient.
// One launch with the logit block resident in LDS between the passes where the shape allows
// (fused_em.hip); otherwise the composition of the kernels above (the second logits read is then
// served by the Infinity Cache for blocks up to ~100 MB).
// ---------------------------------------------------------------------------------------
namespace rlvi {
int try_launch_fused_em(const float *logits, int64_t ld, const int64_t *labels, float *loss_rows, float *pi,
                        int64_t B, int64_t C, float inv_scale, float tol, int maxiter, float *grad,
                        int64_t ldg, float *out, int32_t *out_iters, void *ws, hipStream_t st, int *rc);
}
extern "C" int rlvi_fused_em_f32(const float *logits, int64_t ld, const int64_t *labels,
                                 float *loss_rows, float *pi, int64_t B, int64_t C,
                                 float inv_scale, float tol, int maxiter, float *grad_logits,
                                 int64_t ldg, float *out, int32_t *out_iters, void *ws,
                                 void *stream) {
    if (!loss_rows || !pi) return RLVI_E_NULL;
    // one launch with the block resident in LDS when the shape allows it (fused_em.hip) ...
    if (logits && labels && ws && B > 0 && C > 0 && !(((uintptr_t)labels & 7) || ((uintptr_t)pi & 3) ||
        ((uintptr_t)loss_rows & 3) || ((uintptr_t)out & 3) || ((uintptr_t)ws & 255))) {
        int frc = 0;
        if (rlvi::try_launch_fused_em(logits, ld, labels, loss_rows, pi, B, C, inv_scale, tol, maxiter,
                                      grad_logits, ldg, out, out_iters, ws,
                                      static_cast<hipStream_t>(stream), &frc))
            return frc;
    }
    // ... the composition otherwise
    // 1. l_i = CE(logits_i, y_i) -> loss_rows  (forward only; pi is not used for the scatter)
    // 2. pi <- E-step(l)   (loss_rows becomes l - min l)
    // With `out`: pass 1 leaves its (meaningless: old pi) batch scalars as accumulate records and the E-step
    // launch sweeps them up on the side (rlvi_epoch_end_f32: one of its workgroups reduces and clears the
    // records while it waits for the first totals) -- no finalize launch behind pass 1 (4.7 us of 39 at
    // 65 536 x 100); pass 3 overwrites out.
    int rc = rlvi_mstep_fwd_bwd_f32(logits, ld, labels, nullptr, pi, loss_rows, B, B, C, inv_scale,
                                    nullptr, 0, nullptr, ws, stream);
    if (rc) return rc;
    rc = out != nullptr ? rlvi_epoch_end_f32(loss_rows, pi, B, tol, maxiter, 0, 0.0f, nullptr, 1, out, out_iters, ws, stream)
                        : rlvi_estep_deep_f32(loss_rows, pi, B, tol, maxiter, out_iters, nullptr, ws, stream);
    if (rc) return rc;
    // 3. L = inv_scale * sum pi_i l_i and dL/dlogits with the NEW pi; no scatter
    return rlvi_mstep_fwd_bwd_f32(logits, ld, labels, nullptr, pi, nullptr, B, B, C, inv_scale,
                                  grad_logits, ldg, out, ws, stream);
}

// ---------------------------------------------------------------------------------------
// fp64 X.theta + per-sample NLL.  The contraction runs on the fp64 matrix cores, 16 rows per wave
// (rows16_dot_mfma); a per-row wave dot product (tree sum) is kept as the measured alternative.
// n*d is 15-20 k elements here: launch-latency-bound either way.
// ---------------------------------------------------------------------------------------
namespace rlvi {

// X[row0 .. row0+15] . v on the fp64 matrix cores: the dense X.w / X.theta contraction of the
// linear / logistic paths (standard-learning/rlvi.py:72,:81; online-learning/main.py:295) as
// v_mfma_f64_16x16x4_f64 steps with A = a 16 x 4 panel of X and B = the 4 matching entries of v in every
// column.  Operand maps as in wls.hip: lane l feeds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// the results of a lane are D[row = (l>>4) + 4 r][col = l&15], all columns alike.  Lanes with
// (l&15) < 4 return the dot product of row (l>>4) + 4 (l&15) (`mine` tells which), the others 0.
typedef double aux_d4_t __attribute__((ext_vector_type(4)));
// [kbeg, kend): the columns this call covers (kbeg a multiple of 4; the whole row by default).
__device__ __forceinline__ double rows16_dot_mfma(const double *__restrict__ X, const double *__restrict__ v,
                                                  int64_t row0, int64_t n, int64_t d, int &myrow,
                                                  int64_t kbeg = 0, int64_t kend = -1) {
    if (kend < 0) kend = d;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int64_t row = row0 + i;
    aux_d4_t acc = {0.0, 0.0, 0.0, 0.0};
    const bool rok = row < n;
    const double *xr = X + (rok ? row : 0) * d;
    int64_t k0 = kbeg;
    // four k-panels per trip, their eight loads in flight together (a plain loop waits for one 8-byte load per
    // matrix instruction: 40 -> 17 us for 256 x 561, 12 -> 4.5 us for 256 x 60; eight per trip: no further
    // gain); the accumulation order is the plain loop's: the same bits
    constexpr int XU = 4;
    for (; k0 + 4 * XU <= kend; k0 += 4 * XU) {
        double a[XU], b[XU];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int64_t k = k0 + 4 * u + kk;
            a[u] = rok ? xr[k] : 0.0;
            b[u] = v[k];
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k0 < kend; k0 += 4) {
        const int64_t k = k0 + kk;
        const double a = (rok && k < kend) ? xr[k] : 0.0;
        const double b = k < kend ? v[k] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    myrow = i < 4 ? kk + 4 * i : -1;
    return i == 0 ? acc[0] : i == 1 ? acc[1] : i == 2 ? acc[2] : acc[3];
}

// r_i = (y_i - x_i.theta)^2 -> losses; block partials of {w.r, sum w} -> part
template <bool MFMA>
__global__ __launch_bounds__(256) void linreg_resid_kernel(const double *__restrict__ X,
                                                           const double *__restrict__ y,
                                                           const double *__restrict__ theta,
                                                           const double *__restrict__ w,
                                                           int64_t n, int64_t d,
                                                           double *__restrict__ losses,
                                                           double *__restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double num = 0.0, den = 0.0;
    if (MFMA) {
        for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16; r0 < n; r0 += (int64_t)gridDim.x * 64) {
            int myrow;
            const double p = rows16_dot_mfma(X, theta, r0, n, d, myrow);
            const int64_t i = r0 + myrow;
            if (myrow >= 0 && i < n) {
                const double r = (y[i] - p) * (y[i] - p);
                losses[i] = r;
                num += w[i] * r;
                den += w[i];
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
            double p = 0.0;
            for (int64_t j = lane; j < d; j += 64) p += X[i * d + j] * theta[j];
            p = wave_sum(p);
            const double r = (y[i] - p) * (y[i] - p);
            if (lane == 0) {
                losses[i] = r;
                num += w[i] * r;
                den += w[i];
            }
        }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    __shared__ double sh[8];
    if (lane == 0) { sh[2 * wave] = num; sh[2 * wave + 1] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sh[0] + sh[2] + sh[4] + sh[6];
        part[2 * blockIdx.x + 1] = sh[1] + sh[3] + sh[5] + sh[7];
    }
}

__global__ __launch_bounds__(256) void linreg_scale_kernel(double *__restrict__ losses, int64_t n,
                                                           const double *__restrict__ part,
                                                           int nblocks,
                                                           double *__restrict__ sigma2_out) {
    // every block re-derives sigma2 from the partials in the same fixed order
    double num = 0.0, den = 0.0;
    for (int i = 0; i < nblocks; ++i) { num += part[2 * i]; den += part[2 * i + 1]; }
    const double sigma2 = num / den;
    if (blockIdx.x == 0 && threadIdx.x == 0 && sigma2_out != nullptr) *sigma2_out = sigma2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        losses[i] = 0.5 * losses[i] / sigma2;
}

// -log sigmoid(p) without overflow (online-learning/main.py:295-296 with :84-85 folded in)
__device__ __forceinline__ double logistic_nll_of(double p) {
    return p >= 0.0 ? log1p(exp(-p)) : -p + log1p(exp(p));
}

template <bool MFMA>
__global__ __launch_bounds__(256) void logistic_nll_kernel(const double *__restrict__ X,
                                                           const double *__restrict__ wv, double b,
                                                           int64_t n, int64_t d,
                                                           double *__restrict__ losses) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto nll = [](double p) { return logistic_nll_of(p); };
    if (MFMA) {
        for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16; r0 < n; r0 += (int64_t)gridDim.x * 64) {
            int myrow;
            const double p = rows16_dot_mfma(X, wv, r0, n, d, myrow) + b;
            if (myrow >= 0 && r0 + myrow < n) losses[r0 + myrow] = nll(p);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
            double p = 0.0;
            for (int64_t j = lane; j < d; j += 64) p += X[i * d + j] * wv[j];
            p = wave_sum(p) + b;
            if (lane == 0) losses[i] = nll(p);
        }
    }
}

// Long rows (d >= 128, e.g. the 561 HAR features of online-learning): one workgroup per 16-row block, its
// four waves a quarter of the columns each, the four partial dot products added in wave order -- a quarter of
// the dependent load round trips per wave (256 x 561: 17 -> 6 us).
__global__ __launch_bounds__(256) void logistic_nll_splitk_kernel(const double *__restrict__ X,
                                                                  const double *__restrict__ wv, double b,
                                                                  int64_t n, int64_t d,
                                                                  double *__restrict__ losses) {
    __shared__ double part[4][16];
    const int wave = threadIdx.x >> 6;
    const int64_t dq = ((d + 15) / 16) * 4;                 // columns per wave, a multiple of 4
    const int64_t kbeg = wave * dq, kend = kbeg + dq < d ? kbeg + dq : d;
    for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < n; r0 += (int64_t)gridDim.x * 16) {
        int myrow;
        const double p = kbeg < d ? rows16_dot_mfma(X, wv, r0, n, d, myrow, kbeg, kend) : 0.0;
        if (kbeg >= d) { const int lane = threadIdx.x & 63; myrow = (lane & 15) < 4 ? (lane >> 4) + 4 * (lane & 15) : -1; }
        if (myrow >= 0) part[wave][myrow] = p;
        __syncthreads();
        if (threadIdx.x < 16 && r0 + threadIdx.x < n) {
            const double t = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
            losses[r0 + threadIdx.x] = logistic_nll_of(t + b);
        }
        __syncthreads();
    }
}

}  // namespace rlvi

extern "C" int rlvi_linreg_losses_f64(const double *X, const double *y, const double *theta,
                                      const double *w, int64_t n, int64_t d, double *losses,
                                      double *sigma2_out, void *ws, void *stream) {
    if (!X || !y || !theta || !w || !losses || !ws) return RLVI_E_NULL;
    if (n <= 0 || d <= 0) return RLVI_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // scratch region (NOT the M-step records, which must stay zero between epochs)
    double *part = reinterpret_cast<double *>(static_cast<char *>(ws) + WS_SCRATCH_OFF);
    // X.theta on the fp64 matrix cores (16 rows per wave); RLVI_XW_MFMA=0: one wave dot product per row
    const bool mfma = tune_get("RLVI_XW_MFMA", 1) != 0;
    int nb = (int)(mfma ? (n + 63) / 64 : (n + 3) / 4);
    if (nb > 256) nb = 256;
    int rc = mfma ? launch(linreg_resid_kernel<true>, dim3(nb), dim3(256), 0, st, X, y, theta, w, n, d, losses, part)
                  : launch(linreg_resid_kernel<false>, dim3(nb), dim3(256), 0, st, X, y, theta, w, n, d, losses, part);
    if (rc != 0) return rc;
    int nb2 = (int)((n + 255) / 256);
    if (nb2 > 256) nb2 = 256;
    return launch(linreg_scale_kernel, dim3(nb2), dim3(256), 0, st, losses, n, part, nb, sigma2_out);
}

extern "C" int rlvi_logistic_nll_f64(const double *X, const double *w, double b, int64_t n,
                                     int64_t d, double *losses, void *stream) {
    if (!X || !w || !losses) return RLVI_E_NULL;
    if (n <= 0 || d <= 0) return RLVI_E_SHAPE;
    const bool mfma = tune_get("RLVI_XW_MFMA", 1) != 0;
    int nb = (int)(mfma ? (n + 63) / 64 : (n + 3) / 4);
    if (nb > 1024) nb = 1024;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mfma && d >= 128) {
        int nbk = (int)((n + 15) / 16);
        if (nbk > 4096) nbk = 4096;
        return launch(logistic_nll_splitk_kernel, dim3(nbk), dim3(256), 0, st, X, w, b, n, d, losses);
    }
    return mfma ? launch(logistic_nll_kernel<true>, dim3(nb), dim3(256), 0, st, X, w, b, n, d, losses)
                : launch(logistic_nll_kernel<false>, dim3(nb), dim3(256), 0, st, X, w, b, n, d, losses);
}

// ---------------------------------------------------------------------------------------
// Measurement aid (bench.py's roofline.copy_same_bytes_us): a flat copy, 16 bytes per lane, nontemporal loads
// and stores, one 16-byte piece per lane and trip, grid = 16 waves per CU -- the plainest kernel that moves the
// M-step's bytes (logits in, gradient out), timed in the same rotation and graph as the M-step itself.
// ---------------------------------------------------------------------------------------
namespace rlvi {
typedef unsigned int aux_vu4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(aux_vu4 *__restrict__ dst, const aux_vu4 *__restrict__ src,
                                                          int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
}  // namespace rlvi

extern "C" int rlvi_stream_copy(void *dst, const void *src, size_t bytes, void *stream) {
    if (!dst || !src) return RLVI_E_NULL;
    if (((uintptr_t)dst & 15) || ((uintptr_t)src & 15) || (bytes & 15)) return RLVI_E_ALIGN;
    if (bytes == 0) return 0;
    const int64_t n16 = (int64_t)(bytes / 16);
    int64_t nb = (n16 + 255) / 256;
    const int64_t cap = (int64_t)device_info().cus * 4 * tune_get("RLVI_COPY_WPS", 4);     // 16 waves per CU
    if (nb > cap) nb = cap;
    return launch(stream_copy_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream),
                  static_cast<aux_vu4 *>(dst), static_cast<const aux_vu4 *>(src), n16);
}
