// E-step of the deep-learning variant (train_rlvi.py:14-38) with the fixed point solved as a
// TRAJECTORY instead of a chain of K dependent population-wide reductions.
//
// The reference iterates  r_{k+1} = g(S(r_k)/N),  S(r) = sum_i f(r e_i),  f(t) = t/(1+t),
// e_i = exp(-(l_i - min l)), r_0 = 0.95/(1-0.95), and stops at the first k with
// sqrt(D_k) < tol, D_k = sum_i (f(r_k e_i) - f(r_{k-1} e_i))^2 (D_0 against the caller's pi).
// Every iteration is a reduction over all samples followed by a broadcast: 20 serial exchanges
// (~2.5 us each on MI355X) for ~0.1 us of arithmetic.
//
// Here workgroup (k, s) owns node k of the trajectory and slice s of the samples (registers).
// One ROUND evaluates, for ALL nodes at once,  S(r'_k), dS/dr(r'_k) and D(r'_k, r'_{k-1})  at
// guessed nodes r'_k, exchanges the K*S records once, and then every workgroup runs the scalar
// recurrence with the first-order correction S(r_k) ~ S(r'_k) + S'(r'_k)(r_k - r'_k).  The
// corrected r_k become the next round's nodes.  Node 0 is exact, a node whose predecessor was
// exact becomes exact, so m rounds fix at least m nodes (worst case = the iterative scheme); the
// correction is Newton-like (second-order remainder <= 0.25 (dr/r)^2), so from the previous
// call's trajectory (kept in the workspace) it takes 3 rounds, from a cold geometric guess 5-6.
// A round is accepted when max_k |r_k - r'_k| / r'_k <= 1e-6 over the executed iterations: then
// S is exact to ~1e-12 and D (evaluated AT the nodes) to ~4e-6 relative, i.e. the stop decision
// differs from the reference's only where its own fp32 rounding would decide it.
//
// Sums: fp32 per thread (<= 16 terms), fp64 across threads / workgroups in a fixed order: every
// workgroup computes bit-identical totals and the same scalar recurrence -- deterministic.
#include <stdlib.h>

#include "rlvi_coop.h"

namespace rlvi {

constexpr int TJ_BLOCK = 1024;
constexpr int TJ_NW = TJ_BLOCK / WAVE;
constexpr int TJ_MAXK = 64;
constexpr float TJ_ACCEPT = 1e-6f;

struct TrajState {
    long long n;
    int k;
    int pad;
    float nodes[TJ_MAXK];
};

struct TjShared {
    double part[TJ_NW][3];
    double rec[MAX_COOP_WG][3];
    double tot[TJ_MAXK][3];
    float nodes[2][TJ_MAXK];     // [round parity]: read the old nodes, write the corrected ones
    int dead;
    int res_it;
    float res_delta, res_rfin;
};

// All workgroups publish {a, b, c}; afterwards sh.rec[w][0..2] holds every workgroup's record.
// Same protocol as rlvi_coop.h (self-tagged 8-byte granules, sc1 stores / loads, parity slots).
__device__ __forceinline__ void tj_exchange(TjShared &sh, double a, double b, double c, gu64 *slots,
                                            uint32_t tag, int step, int nwg, int32_t *status,
                                            bool &dead) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    a = group_allreduce<WAVE>(a, FAdd());
    b = group_allreduce<WAVE>(b, FAdd());
    c = group_allreduce<WAVE>(c, FAdd());
    if (lane == 0) { sh.part[wave][0] = a; sh.part[wave][1] = b; sh.part[wave][2] = c; }
    __syncthreads();
    if (wave == 0) {
        double t[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double v = lane < TJ_NW ? sh.part[lane][q] : 0.0;
            t[q] = group_allreduce<WAVE>(v, FAdd());
        }
        gu64 *buf = slots + (size_t)(step & 1) * MAX_COOP_WG * XCHG2_GRANULES;
        if (!dead) {
            if (lane < 6) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(t[lane >> 1]);
                const uint32_t half = (lane & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
                __hip_atomic_store(buf + (size_t)blockIdx.x * XCHG2_GRANULES + lane,
                                   ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long t0 = wall_clock64();
            bool timeout = false;
            // lane l owns records l, l+64, l+128, l+192: all of them are polled together
            constexpr int PER = MAX_COOP_WG / WAVE;
            unsigned long long x[PER][6];
            for (unsigned spin = 0;; ++spin) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int w = lane + u * WAVE;
                    if (w < nwg) {
                        gu64 *p = buf + (size_t)w * XCHG2_GRANULES;
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            x[u][q] = __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int q = 0; q < 6; ++q) ok = ok && (uint32_t)(x[u][q] >> 32) == tag;
                    }
                }
                if (__all(ok)) break;
                if ((spin & 63u) == 63u && wall_clock64() - t0 > SPIN_BOUND_TICKS) {
                    timeout = true;
                    break;
                }
            }
            if (!timeout) {
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int w = lane + u * WAVE;
                    if (w < nwg) {
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            sh.rec[w][q] = __longlong_as_double((long long)(
                                ((x[u][2 * q + 1] & 0xFFFFFFFFull) << 32) | (x[u][2 * q] & 0xFFFFFFFFull)));
                    }
                }
            }
            if (__any(timeout)) {
                dead = true;
                if (lane == 0) atomicOr(status, RLVI_ST_TIMEOUT);
            }
        }
        if (lane == 0) sh.dead = dead ? 1 : 0;
    }
    __syncthreads();
    dead = sh.dead != 0;
}

// E > 0: the slice lives in registers (slice length <= 1024*E); E == 0: re-read from memory.
template <int E>
__global__ __launch_bounds__(TJ_BLOCK) void estep_traj_kernel(
    float *__restrict__ res, float *__restrict__ wts, int64_t N, float tol, int K, int S,
    int32_t *__restrict__ out_iters, float *__restrict__ trace, void *ws,
    float *__restrict__ mstep_out, double mstep_scale, unsigned long long *__restrict__ dbg) {
    const int nwg = K * S;
    int dbgi = 0;
#define TJ_STAMP() do { if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && dbgi < 60) dbg[dbgi++] = wall_clock64(); } while (0)
    TJ_STAMP();
    if ((int)blockIdx.x == nwg) {   // epoch end: reduce + clear the M-step records (own CU)
        double *part = reinterpret_cast<double *>(static_cast<char *>(ws) + WS_PART_OFF);
        reduce_partials(part, MSTEP_MAX_BLOCKS, mstep_scale, mstep_out, true, TJ_BLOCK);
        return;
    }
    __shared__ TjShared sh;
    char *wsb = static_cast<char *>(ws);
    WsHeader *hdr = reinterpret_cast<WsHeader *>(wsb);
    gu64 *slots = (gu64 *)(reinterpret_cast<unsigned long long *>(wsb + WS_XCHG2_OFF));
    TrajState *state = reinterpret_cast<TrajState *>(wsb + WS_TRAJ_OFF);
    uint32_t tag = __hip_atomic_load((gu32 *)&hdr->epoch_base, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) + 1u;
    int xstep = 0;
    bool dead = false;

    const int tid = threadIdx.x;
    const int k = (int)blockIdx.x / S, s = (int)blockIdx.x - k * S;
    const int64_t L = (N + S - 1) / S;
    const int64_t lo = (int64_t)s * L;
    const int64_t hi = lo + L < N ? lo + L : N;
    constexpr int EE = E > 0 ? E : 1;

    // ---- slice -> registers, local min
    float e[EE], w0[EE];
    int cnt = 0;
    float mn = __builtin_inff();
    if (E > 0) {
#pragma unroll
        for (int j = 0; j < EE; ++j) {
            const int64_t i = lo + tid + (int64_t)j * TJ_BLOCK;
            const bool ok = i < hi;
            e[j] = ok ? res[i] : __builtin_inff();
            w0[j] = (ok && k == 0) ? wts[i] : 0.0f;     // caller's pi: only D_0 needs it
            if (ok) { cnt = j + 1; mn = fminf(mn, e[j]); }
        }
    } else {
        for (int64_t i = lo + tid; i < hi; i += TJ_BLOCK) mn = fminf(mn, res[i]);
    }
    {
        // min via the sum exchange: every workgroup gets all local minima and takes their min
        double a = (double)group_allreduce<WAVE>(mn, FMin());
        const int lane = tid & 63, wave = tid >> 6;
        __shared__ float smin[TJ_NW];
        if (lane == 0) smin[wave] = (float)a;
        __syncthreads();
        float bm = smin[0];
#pragma unroll
        for (int w = 1; w < TJ_NW; ++w) bm = fminf(bm, smin[w]);
        // publish the block minimum once (thread 0's value after the wave butterfly: divide by 64
        // lanes * 16 waves is avoided by sending it in slot a of lane 0 only)
        double pa = (tid == 0) ? (double)bm : 0.0;
        tj_exchange(sh, pa, 0.0, 0.0, slots, tag, xstep, nwg, &hdr->status, dead);
        ++tag; ++xstep;
        float g = __builtin_inff();
        for (int w = tid & 63; w < nwg; w += WAVE) g = fminf(g, (float)sh.rec[w][0]);
        mn = group_allreduce<WAVE>(g, FMin());
    }
    TJ_STAMP();   // after the min exchange
    // residuals.sub_(min) (:27), e = exp(-residuals) (:28); node-0 workgroups store the shift
    if (E > 0) {
#pragma unroll
        for (int j = 0; j < EE; ++j) {
            const int64_t i = lo + tid + (int64_t)j * TJ_BLOCK;
            const float l = e[j] - mn;
            if (j < cnt && k == 0) res[i] = l;
            e[j] = j < cnt ? expf(-l) : 0.0f;
        }
    }
    // (streaming form: the slice is re-read unshifted every round; the shift is stored in the final
    //  phase by the one workgroup that owns the element there, after everybody's last read)

    // ---- initial nodes: last call's trajectory if it was for the same N and K, else geometric
    const bool warm = state->n == (long long)N && state->k == K;
    if (tid < K) sh.nodes[0][tid] = warm ? state->nodes[tid] : 19.0f * exp2f(-(float)tid);
    __syncthreads();
    float r_mine = sh.nodes[0][k];
    float r_prev = k > 0 ? sh.nodes[0][k - 1] : 0.0f;
    if (k == 0) r_mine = (float)(0.95 / (1.0 - 0.95));
    __syncthreads();

    const float invN = 1.0f / (float)N;
    int it = K;
    float r_fin = r_mine;
    int cur = 0;
    for (int round = 0; round <= K; ++round) {
        // ---- sums of this workgroup's node over its slice
        float fS = 0.0f, fP = 0.0f, fD = 0.0f;
        auto body = [&](float ev, float wv) {
            const float t = r_mine * ev;
            const float inv = __builtin_amdgcn_rcpf(1.0f + t);
            const float f = t * inv;
            float fp;
            if (k == 0) fp = wv;
            else { const float tp = r_prev * ev; fp = tp * __builtin_amdgcn_rcpf(1.0f + tp); }
            const float d = f - fp;
            fS += f;
            fP += ev * inv * inv;
            fD += d * d;
        };
        if (E > 0) {
#pragma unroll
            for (int j = 0; j < EE; ++j)
                if (j < cnt) body(e[j], w0[j]);
        } else {
            for (int64_t i = lo + tid; i < hi; i += TJ_BLOCK)
                body(expf(-(res[i] - mn)), k == 0 ? wts[i] : 0.0f);
        }
        TJ_STAMP();   // sums done
        tj_exchange(sh, (double)fS, (double)fP, (double)fD, slots, tag, xstep, nwg, &hdr->status, dead);
        ++tag; ++xstep;
        TJ_STAMP();   // exchange done
        // ---- totals per node, fixed order over the slices
        if (tid < K) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                double t = 0.0;
                for (int ss = 0; ss < S; ++ss) t += sh.rec[tid * S + ss][q];
                sh.tot[tid][q] = t;
            }
        }
        __syncthreads();
        // ---- scalar recurrence with the first-order correction: wave 0 only (redundant copies
        // on all 16 waves would share the CU's issue slots); lane j holds node j's data, the
        // serial chain reads it with v_readlane, everything else (errors, stop index, delta) is
        // lane-parallel
        if (tid < WAVE) {
            const int lane = tid;
            const bool has = lane < K;
            const float rn_l = has ? sh.nodes[cur][lane] : 1.0f;
            // lane-parallel: a0 = mean(pi) at the node, b = d mean / dr (fp32 from the fp64 totals)
            const float a0_l = has ? (float)sh.tot[lane][0] * invN : 0.0f;          // (:35)
            const float b_l = has ? (float)(sh.tot[lane][1] * (double)invN) : 0.0f;
            const float err_l = has ? (float)sqrt(sh.tot[lane][2]) : __builtin_inff();   // (:33)
            const unsigned long long stopmask = __ballot(has && err_l < tol);           // (:36)
            const int it_now = stopmask ? (int)__builtin_ctzll(stopmask) + 1 : K;
            const int steps = it_now + 2 < K ? it_now + 2 : K;      // a little lookahead
            float r = (float)(0.95 / (1.0 - 0.95));
            float rnew_l = rn_l;                                     // corrected node of this lane
            float avg_l = 0.0f;
            // serial chain, fully unrolled so that `step` is a literal (v_readlane, no LDS):
            // avg = a0 + b (r - r'), r <- avg / (1 - avg): five dependent fp32 operations per step
#pragma unroll
            for (int step = 0; step < TJ_MAXK; ++step) {
                if (step < steps) {
                    const float rn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rn_l), step));
                    const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0_l), step));
                    const float bb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), step));
                    if (lane == step) rnew_l = r;
                    // damped step: the linearisation is trusted within +-50% of the node (S is
                    // concave in r, so the extrapolated mean stays positive); mean(pi) < 1
                    const float h = 0.5f * rn;
                    const float d = fmaxf(fminf(r - rn, h), -h);
                    const float avg = fminf(fmaf(bb, d, a0), 0.999999f);
                    if (lane == step) avg_l = avg;
                    r = fmaxf(avg * __builtin_amdgcn_rcpf(1.0f - avg), 1e-37f);   // (:31)
                }
            }
            // nodes beyond the lookahead: a fresh geometric tail from the last corrected node
            if (has && lane >= steps) {
                const float last = __shfl(rnew_l, steps - 1, WAVE);
                rnew_l = last * exp2f(-(float)(lane - steps + 1));
            }
            float d_l = (has && lane < it_now) ? fabsf(rnew_l - rn_l) * __builtin_amdgcn_rcpf(rn_l) : 0.0f;
            float delta_w = group_allreduce<WAVE>(d_l, FMax());
            // Early accept: with nodes off by delta the corrected r are good to 0.25 delta^2, and
            // the errors (evaluated AT the nodes) to about delta*(r_k + r_{k-1})/|r_k - r_{k-1}|
            // relative.  If every stop test up to `it` clears tol by 8x that margin, the stop
            // index cannot change any more, and neither can pi: no verification round needed.
            // (Not when the caller asked for the error trace: that wants the errors themselves.)
            if (delta_w > TJ_ACCEPT && delta_w <= 1e-3f && trace == nullptr) {
                const float rp_l = __shfl_up(rn_l, 1, WAVE);
                float u = 0.0f;
                if (has && lane < it_now && lane > 0) {
                    const float gap = fabsf(rn_l - rp_l);
                    u = 8.0f * delta_w * (rn_l + rp_l) * __builtin_amdgcn_rcpf(fmaxf(gap, 1e-30f));
                }
                if (lane == 0) u = 128.0f * delta_w;       // D_0 is taken against the caller's pi
                const bool unsafe = has && lane < it_now && fabsf(err_l - tol) <= u * err_l;
                if (__ballot(unsafe) == 0ull) delta_w = 0.0f;
            }
            if (has) sh.nodes[cur ^ 1][lane] = rnew_l;
            if (trace != nullptr && blockIdx.x == 0 && lane < it_now) {
                trace[2 * lane] = err_l;
                trace[2 * lane + 1] = avg_l;
            }
            const float rfin_w = __shfl(rnew_l, it_now - 1, WAVE);   // all lanes take part
            if (lane == 0) {
                sh.res_it = it_now;
                sh.res_delta = delta_w;
                sh.res_rfin = rfin_w;
            }
        }
        __syncthreads();
        it = sh.res_it;
        r_fin = sh.res_rfin;
        const float delta = sh.res_delta;
        r_mine = (k == 0) ? (float)(0.95 / (1.0 - 0.95)) : sh.nodes[cur ^ 1][k];
        r_prev = k > 0 ? sh.nodes[cur ^ 1][k - 1] : 0.0f;
        TJ_STAMP();   // recurrence done
        cur ^= 1;
        __syncthreads();          // sh.tot / the other nodes buffer are rewritten next round
        if (delta <= TJ_ACCEPT || dead) break;
    }

    // ---- weights = pi / max(pi); max is attained at e = 1 (the min-residual sample) (:38)
    const float tmax = r_fin;
    const float inv_pmax = (1.0f + tmax) * __builtin_amdgcn_rcpf(tmax);
    // workgroup (k, s) writes the elements of slice s whose position is congruent to k mod K
    // (position mod K advances by 1024 mod K per step: no per-element division)
    {
        const int stepm = TJ_BLOCK % K;
        int pm = tid % K;
        if (E > 0) {
#pragma unroll
            for (int j = 0; j < EE; ++j) {
                if (j < cnt && pm == k) {
                    const int64_t i = lo + tid + (int64_t)j * TJ_BLOCK;
                    const float t = r_fin * e[j];
                    wts[i] = t * __builtin_amdgcn_rcpf(1.0f + t) * inv_pmax;
                }
                pm += stepm;
                pm = pm >= K ? pm - K : pm;
            }
        } else {
            for (int64_t i = lo + tid; i < hi; i += TJ_BLOCK) {
                if (pm == k) {
                    const float l = res[i] - mn;
                    const float t = r_fin * expf(-l);
                    res[i] = l;
                    wts[i] = t * __builtin_amdgcn_rcpf(1.0f + t) * inv_pmax;
                }
                pm += stepm;
                pm = pm >= K ? pm - K : pm;
            }
        }
    }
    TJ_STAMP();   // final stores issued
    if (dbg != nullptr && blockIdx.x == 0 && tid == 0) dbg[63] = (unsigned long long)dbgi;
    if (blockIdx.x == 0 && tid == 0) {
        if (out_iters != nullptr) *out_iters = it;
        state->n = (long long)N;
        state->k = K;
        for (int q = 0; q < K; ++q) state->nodes[q] = sh.nodes[cur][q];
        __hip_atomic_store((gu32 *)&hdr->epoch_base, tag + 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Eligibility + geometry.  Returns 1 if launched (rc in *rc), 0 if the caller should use the
// iterative kernel.
int try_launch_estep_traj(float *res, float *wts, int64_t N, float tol, int maxiter,
                          int32_t *out_iters, float *trace, void *ws, hipStream_t st,
                          float *mstep_out, double mstep_scale, int *rc) {
    static const int mode = getenv("RLVI_ESTEP_TRAJ") ? atoi(getenv("RLVI_ESTEP_TRAJ")) : 1;
    if (mode == 0 || maxiter < 1 || maxiter > TJ_MAXK || N < 4096) return 0;
    const int K = maxiter;
    int S = (MAX_COOP_WG - 1) / K;
    if (S < 1) return 0;
    static const int force_s = getenv("RLVI_TJ_S") ? atoi(getenv("RLVI_TJ_S")) : 0;
    if (force_s > 0 && force_s < S) S = force_s;
    const int64_t smax = (N + 4095) / 4096;        // at least 4 samples per thread and slice
    if (S > smax) S = (int)smax;
    const int64_t L = (N + S - 1) / S;
    const unsigned grid = (unsigned)(K * S) + (mstep_out != nullptr ? 1u : 0u);
    static const int debug = getenv("RLVI_TJ_DEBUG") ? atoi(getenv("RLVI_TJ_DEBUG")) : 0;
    unsigned long long *dbg = debug ? reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF) : nullptr;
#define RLVI_TJ(E_)                                                                              \
    hipLaunchKernelGGL((estep_traj_kernel<E_>), dim3(grid), dim3(TJ_BLOCK), 0, st, res, wts, N,  \
                       tol, K, S, out_iters, trace, ws, mstep_out, mstep_scale, dbg)
    if (L <= (int64_t)TJ_BLOCK * 4) RLVI_TJ(4);
    else if (L <= (int64_t)TJ_BLOCK * 16) RLVI_TJ(16);
    else if (L <= (int64_t)TJ_BLOCK * 64) RLVI_TJ(64);
    else RLVI_TJ(0);
#undef RLVI_TJ
    *rc = (int)hipGetLastError();
    return 1;
}

}  // namespace rlvi
