// E-step fixed point (latent-Bernoulli posteriors) in one cooperative launch.
//
//   VAR_DEEP    update_sample_weights   deep-learning/methods/train_rlvi.py:14-38   (fp32)
//   VAR_STD     update_weights          standard-learning/rlvi.py:8-20              (fp64)
//   VAR_ONLINE  update_weights_rlvi     online-learning/main.py:45-58               (fp64)
//
// Each of G workgroups (256 or 1024 threads; G is checked against what the occupancy query says is
// co-resident on this device) keeps E elements per thread of e_i=exp(-l_i)
// and pi_i in registers for the whole fixed point; per iteration the only traffic is the
// 32-byte record exchange of rlvi_coop.h ({sum pi, sum (pi'-pi)^2} as doubles).  Sums are
// accumulated in fp64 in a fixed order, so the result is deterministic and every workgroup takes
// the same stop decision.  Latency-bound: N*4 B <= a few MB; report us and iterations, not GB/s.
#include <stdlib.h>

#include "rlvi_coop.h"

namespace rlvi {

enum { VAR_DEEP = 0, VAR_STD = 1, VAR_ONLINE = 2 };

// estep_trajb.hip: the trajectory solver of the deep variant (returns 0 when not applicable)
int try_launch_estep_trajb(float *res, float *wts, int64_t N, float tol, int maxiter,
                           int32_t *out_iters, float *trace, void *ws, hipStream_t st,
                           float *mstep_out, double mstep_scale, int *rc, int64_t n_all = 0,
                           int sharded = 0, int dry_run = 0);
int peers_world_of(const void *ws);     // peer.hip: 0 = rlvi_workspace_set_peers never ran on it

// t/(1+t) etc.: fp32 uses v_rcp_f32 (1 ulp) + multiply instead of the ~15-instruction IEEE
// division sequence: <= 2 ulp on pi, two orders of magnitude inside the 1e-5 parity budget, and
// the division is the bulk of the per-iteration arithmetic.  fp64 keeps the exact division.
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double fdiv(double a, double b) { return a / b; }

template <typename F>
__device__ __forceinline__ F fexp(F x);
template <>
__device__ __forceinline__ float fexp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double fexp<double>(double x) { return exp(x); }

// in : deep: residuals (in/out, min-shifted), weights (in/out)
//      std/online: losses (read only), out (write only)
template <typename F, int VAR, int E, int ESTEP_BLOCK>
__global__ __launch_bounds__(ESTEP_BLOCK) void estep_kernel(F *__restrict__ res,
                                                            F *__restrict__ wts, int64_t N,
                                                            F tol, int maxiter,
                                                            int32_t *__restrict__ out_iters,
                                                            F *__restrict__ trace, void *ws,
                                                            float *__restrict__ mstep_out,
                                                            double mstep_scale) {
    // epoch end: one extra workgroup (the last block, on a CU of its own) reduces and clears the
    // M-step partial records of the epoch while the others run the fixed point
    const int nwg = (int)gridDim.x - (mstep_out != nullptr ? 1 : 0);
    if ((int)blockIdx.x == nwg) {
        double *part = reinterpret_cast<double *>(static_cast<char *>(ws) + WS_PART_OFF);
        reduce_partials(part, MSTEP_MAX_BLOCKS, mstep_scale, mstep_out, true, ESTEP_BLOCK);
        return;
    }
    Coop<ESTEP_BLOCK> co;
    co.init(ws, nwg);
    const int64_t gstride = (int64_t)nwg * ESTEP_BLOCK;
    const int64_t i0 = (int64_t)blockIdx.x * ESTEP_BLOCK + threadIdx.x;

    // this thread's elements are i0 + j*gstride, j < cnt (a prefix of 0..E-1)
    int cnt = 0;
    if (i0 < N) {
        const int64_t c = (N - i0 + gstride - 1) / gstride;
        cnt = c < E ? (int)c : E;
    }
#define ok_(j) ((j) < cnt)
    F e[E], w[E];
    F mn = (F)__builtin_inf();
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = i0 + j * gstride;
        e[j] = ok_(j) ? res[i] : (F)0;
        if (VAR == VAR_DEEP) {
            w[j] = ok_(j) ? wts[i] : (F)0;          // caller's pi: enters the first error only (:33)
            if (ok_(j)) mn = e[j] < mn ? e[j] : mn;
        } else {
            w[j] = ok_(j) ? (F)(VAR == VAR_STD ? 0.95 : 0.5) : (F)0;   // rlvi.py:10 / main.py:48
        }
    }
    if (VAR == VAR_DEEP) {
        double a = (double)mn, b = 0.0;
        co.template allreduce2<OpMin, OpSum>(a, b);   // residuals.min()  (:27)
        mn = (F)a;
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (VAR == VAR_DEEP) {
            const F l = e[j] - mn;                    // residuals.sub_(min)  (:27)
            if (ok_(j)) res[i0 + j * gstride] = l;
            e[j] = ok_(j) ? fexp<F>(-l) : (F)0;        // exp(-residuals)      (:28)
        } else {
            e[j] = ok_(j) ? fexp<F>(-e[j]) : (F)0;     // exp(-losses)
        }
    }

    // first ratio: deep 0.95/(1-0.95) as a python float cast to fp32 (=19.0f);
    // std eps=1-0.95, ratio=eps/(1-eps); online avg=0.5, ratio=1
    F ratio;
    if (VAR == VAR_DEEP) ratio = (F)(0.95 / (1.0 - 0.95));
    else if (VAR == VAR_STD) { const double eps = 1.0 - 0.95; ratio = (F)(eps / (1.0 - eps)); }
    else ratio = (F)(0.5 / (1.0 - 0.5));

    int it = 0;
    F ratio_used = ratio;     // the ratio the latest pi was computed with
    while (it < maxiter) {
        ratio_used = ratio;
        // per-thread partial sums in the working precision (E <= 32 terms), fp64 across threads
        F sse_t = (F)0, sum_t = (F)0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            F nw;
            if (VAR == VAR_STD) {
                nw = fdiv(e[j], ratio + e[j]);        // rlvi.py:15
            } else {
                const F t = ratio * e[j];             // train_rlvi.py:32 / main.py:52
                nw = fdiv(t, (F)1 + t);
            }
            nw = ok_(j) ? nw : (F)0;
            const F d = nw - w[j];
            sse_t += d * d;
            sum_t += nw;
            w[j] = nw;
        }
        double sse = (double)sse_t, sum = (double)sum_t;
        co.template allreduce2<OpSum, OpSum, sizeof(F) == 4>(sse, sum);
        const F err = sizeof(F) == 4 ? (F)sqrtf((float)sse) : (F)sqrt(sse);   // ||new - weights||_2
        const F avg = (F)sum / (F)N;                  // mean(weights)
        if (trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
            trace[2 * it] = err;
            trace[2 * it + 1] = avg;
        }
        ++it;
        if (err < tol) break;
        if (VAR == VAR_STD) {
            const F eps = (F)1 - avg;                 // rlvi.py:13-14
            ratio = eps / ((F)1 - eps);
        } else {
            ratio = avg / ((F)1 - avg);               // train_rlvi.py:31 / main.py:51
        }
    }

    if (VAR != VAR_STD) {
        F mx = -(F)__builtin_inf();
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (ok_(j)) mx = w[j] > mx ? w[j] : mx;
        if (VAR == VAR_DEEP && it > 0) {
            // pi is increasing in e and the min-shifted residual 0 gives e = 1 exactly, so the
            // maximum is the value every thread can compute alone -- no exchange
            const F t1 = ratio_used * (F)1;
            mx = fdiv(t1, (F)1 + t1);     // the very expression used per element: pi_max / mx == 1
        } else {
            double a = (double)mx, b = 0.0;
            co.template allreduce2<OpMax, OpSum>(a, b);
            mx = (F)a;
        }
        if (VAR == VAR_ONLINE) mx = mx * (F)N;        // new /= max(new)*len(new)  (main.py:57)
        // (a wait that timed out -- RLVI_ST_TIMEOUT -- leaves the output as it was: the host raises
        //  on the status, it never hands out sums of a partial population)
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (ok_(j) && !co.dead) wts[i0 + j * gstride] = w[j] / mx;   // weights.div_(max)   (:38)
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (ok_(j) && !co.dead) wts[i0 + j * gstride] = w[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (out_iters != nullptr) *out_iters = it;
        co.finish(ws);
    }
#undef ok_
}

// Geometry: per iteration a workgroup pays (i) E elements per thread of arithmetic, (ii) a
// two-stage fp64 reduction whose second stage grows with the wave count, (iii) one record exchange.
// 256-thread workgroups (4 waves) with E = 8 (2048 samples per CU) measured fastest up to 63 groups; larger
// vectors fall back to 1024-thread workgroups so that G stays within one workgroup per CU.
template <typename F, int VAR>
static int launch_estep(F *res, F *wts, int64_t N, F tol, int maxiter, int32_t *out_iters,
                        F *trace, void *ws, hipStream_t st, float *mstep_out = nullptr,
                        double mstep_scale = 1.0) {
    if constexpr (VAR == VAR_DEEP) {
        int rc = 0;
        if (try_launch_estep_trajb(res, wts, N, tol, maxiter, out_iters, trace, ws, st, mstep_out,
                                   mstep_scale, &rc))
            return rc;
    }
    const int force_e = tune_get("RLVI_ESTEP_E", 0);
    const int force_b = tune_get("RLVI_ESTEP_BLOCK", 0);
    auto groups = [&](int64_t blk, int e) { return (N + blk * e - 1) / (blk * e); };
    const int extra = mstep_out != nullptr ? 1 : 0;
    // every geometry is admitted only if all its workgroups (+ the epoch-end reduction workgroup)
    // are provably co-resident on this device (occupancy query x CUs) and fit the exchange slots
#define RLVI_LAUNCH(E_, B_, LIM_)                                                                \
    do {                                                                                         \
        auto kern = estep_kernel<F, VAR, E_, B_>;                                                \
        const int64_t g_ = groups(B_, E_);                                                       \
        int cap_ = coop_cap(kern, B_) - extra;                                                   \
        if (cap_ > (LIM_)) cap_ = (LIM_);                                                        \
        if (g_ <= cap_)                                                                          \
            return launch(kern, dim3((unsigned)(g_ + extra)), dim3(B_), 0, st, res, wts, N, tol, \
                          maxiter, out_iters, trace, ws, mstep_out, mstep_scale);                \
    } while (0)
    const int lim = MAX_COOP_WG - 1;    // exchange slots, one kept for the epoch-end reduction workgroup
    if (force_b == 256 || force_b == 0) {
        if (force_e == 8 || !force_e) RLVI_LAUNCH(8, 256, lim / 4);
        if (force_e == 16 || !force_e) RLVI_LAUNCH(16, 256, lim / 4);
    }
    if (force_e == 4) RLVI_LAUNCH(4, 1024, lim);
    RLVI_LAUNCH(8, 1024, lim);
    RLVI_LAUNCH(16, 1024, lim);
    if constexpr (sizeof(F) == 4) {
        RLVI_LAUNCH(32, 1024, lim);
    }
#undef RLVI_LAUNCH
    return RLVI_E_LIMIT;
}

}  // namespace rlvi

using namespace rlvi;

extern "C" int rlvi_estep_deep_f32(float *residuals, float *weights, int64_t N, float tol,
                                   int maxiter, int32_t *out_iters, float *trace, void *ws,
                                   void *stream) {
    if (!residuals || !weights || !ws) return RLVI_E_NULL;
    if (N <= 0 || maxiter < 0) return RLVI_E_SHAPE;
    if (((uintptr_t)residuals & 3) || ((uintptr_t)weights & 3) || ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    return launch_estep<float, VAR_DEEP>(residuals, weights, N, tol, maxiter, out_iters, trace, ws,
                                         static_cast<hipStream_t>(stream));
}

// The E-step with the samples sharded over the ranks of one node (one process per GPU): this rank's
// n_local residuals / weights, n_all samples over all ranks.  Only the per-node totals of the trajectory
// solve cross the GPUs (a few hundred bytes per rank and round, written by the kernel itself into the
// peers' inboxes: rlvi_workspace_set_peers); every rank ends with the same fixed point and its own
// slice of pi.  RLVI_E_LIMIT if the trajectory kernel does not take this shape (then gather the
// residuals and run rlvi_estep_deep_f32 on the whole vector).
// out != NULL: also this rank's M-step scalars of the epoch, as rlvi_epoch_end_f32 reduces them
// (scaled by 1/batches) -- the epoch end of a rank in one launch.
extern "C" int rlvi_estep_sharded_f32(float *residuals, float *weights, int64_t n_local, int64_t n_all,
                                      float tol, int maxiter, int64_t batches, float *out,
                                      int32_t *out_iters, void *ws, void *stream) {
    if (!residuals || !weights || !ws) return RLVI_E_NULL;
    if (n_local <= 0 || n_all < n_local || maxiter < 0 || batches < 0) return RLVI_E_SHAPE;
    if (((uintptr_t)residuals & 3) || ((uintptr_t)weights & 3) || ((uintptr_t)out & 3) || ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    if (peers_world_of(ws) < 1) return RLVI_E_WS;      // no peer table in this workspace
    int rc = 0;
    if (try_launch_estep_trajb(residuals, weights, n_local, tol, maxiter, out_iters, nullptr, ws,
                               static_cast<hipStream_t>(stream), out, batches > 0 ? 1.0 / (double)batches : 1.0,
                               &rc, n_all, 1))
        return rc;
    return RLVI_E_LIMIT;
}

// Would rlvi_estep_sharded_f32 launch for this shape on this device, now (with the co-residency this
// process is entitled to)?  0 = yes, RLVI_E_LIMIT = no.  Nothing is launched (occupancy queries only): every rank asks
// before the first collective call and the ranks compare answers (rlvi_amd.dist.set_owner_sharding,
// bench.py), so that nobody starts a solve a peer cannot join.
extern "C" int rlvi_estep_sharded_check(int64_t n_local, int64_t n_all, int maxiter, int with_out) {
    if (n_local <= 0 || n_all < n_local || maxiter < 0) return RLVI_E_SHAPE;
    int rc = 0;
    float dummy_out = 0.0f;
    if (try_launch_estep_trajb(nullptr, nullptr, n_local, 1e-3f, maxiter, nullptr, nullptr, nullptr, nullptr,
                               with_out ? &dummy_out : nullptr, 1.0, &rc, n_all, 1, 1))
        return 0;
    return RLVI_E_LIMIT;
}

extern "C" int rlvi_update_weights_f64(const double *losses, int64_t n, double tol, int maxiter,
                                       double *out, int32_t *out_iters, void *ws, void *stream) {
    if (!losses || !out || !ws) return RLVI_E_NULL;
    if (n <= 0 || maxiter < 0) return RLVI_E_SHAPE;
    if (((uintptr_t)losses & 7) || ((uintptr_t)out & 7) || ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    return launch_estep<double, VAR_STD>(const_cast<double *>(losses), out, n, tol, maxiter,
                                         out_iters, nullptr, ws, static_cast<hipStream_t>(stream));
}

extern "C" int rlvi_update_weights_online_f64(const double *losses, int64_t n, double tol,
                                              int maxiter, double *out, int32_t *out_iters,
                                              void *ws, void *stream) {
    if (!losses || !out || !ws) return RLVI_E_NULL;
    if (n <= 0 || maxiter < 0) return RLVI_E_SHAPE;
    if (((uintptr_t)losses & 7) || ((uintptr_t)out & 7) || ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    return launch_estep<double, VAR_ONLINE>(const_cast<double *>(losses), out, n, tol, maxiter,
                                            out_iters, nullptr, ws,
                                            static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------
// End of a train_rlvi epoch (train_rlvi.py:99-105): E-step over all N samples, optional
// truncation, and the epoch's M-step scalars (sum over the accumulate-mode M-step calls since
// the last epoch end, scaled by 1/batches: out[1] is the reference's train_acc in percent).
// ---------------------------------------------------------------------------------------
extern "C" int rlvi_epoch_end_f32(float *residuals, float *weights, int64_t N, float tol,
                                  int maxiter, int overfit, float alpha, float *thr_inout,
                                  int64_t batches, float *out, int32_t *out_iters, void *ws,
                                  void *stream) {
    if (!residuals || !weights || !ws || (overfit && !thr_inout)) return RLVI_E_NULL;
    if (N <= 0 || maxiter < 0 || batches < 0) return RLVI_E_SHAPE;
    if (((uintptr_t)residuals & 3) || ((uintptr_t)weights & 3) || ((uintptr_t)ws & 255))
        return RLVI_E_ALIGN;
    int rc = launch_estep<float, VAR_DEEP>(residuals, weights, N, tol, maxiter, out_iters, nullptr,
                                           ws, static_cast<hipStream_t>(stream), out,
                                           batches > 0 ? 1.0 / (double)batches : 1.0);
    if (rc || !overfit) return rc;
    return rlvi_threshold_truncate_f32(weights, N, alpha, thr_inout, nullptr, nullptr, ws, stream);
}
