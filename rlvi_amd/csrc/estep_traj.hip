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
// correction is Newton-like (second-order remainder <= 0.25 (dr/r)^2) inside a 25 % trust region;
// outside it the recurrence runs on a global Hermite model of mean(pi) over log r built from ALL
// evaluated nodes (rlvi_traj.h), so from the previous call's trajectory (kept in the workspace) it
// takes 2 rounds and from a cold or poor guess 3.  A round is accepted when max_k |r_k - r'_k| / r'_k <= 1e-6 over the
// executed iterations (S exact to ~1e-12, D -- evaluated AT the nodes -- to ~4e-6 relative), or
// earlier when every stop test clears tol by 8x the uncertainty the remaining delta implies.
// The minimum residual is not exchanged separately: round 0 evaluates with the previous call's
// minimum as the shift (e' = e * exp(shift - min) only rescales the nodes) and carries the true
// minimum in its records.
//
// Sums: fp32 per thread and wave, fp64 across waves / workgroups in a fixed order: every workgroup
// computes bit-identical totals and the same scalar recurrence.  Identical inputs AND workspace
// state give identical bits; a different warm start may move pi by one ulp.
// RLVI_TJ_DEBUG=1 makes workgroup 0 write wall-clock stamps of its phases into the workspace
// scratch (tools/time_parts.py prints them); the stamps never feed a result.
#include <stdlib.h>

#include "rlvi_traj.h"

namespace rlvi {

typedef unsigned int vu4 __attribute__((ext_vector_type(4)));

constexpr int TJ_BLOCK = 1024;
constexpr int TJ_NW = TJ_BLOCK / WAVE;
constexpr int TJ_MAXS = 8;       // slices per node (one lane of the polling wave sweeps them)

struct TjShared {
    double part[TJ_NW][4];
    double rec[MAX_COOP_WG][5];   // gathered records {S, S', Q, D, min} (wave 0 only)
    float pmin[TJ_NW];
    TjOut out;
};

// ---------------------------------------------------------------------------------------
// One round's communication + the scalar recurrence.  Workgroup b = k*S + s publishes
// {S (fp64: two granules), S', Q = -S''/2, D, min (fp32: one each)} as self-tagged granules; waves 0..3 sweep the K*S records (one 64-byte record
// per lane, four 16-byte sc1 loads in flight) into LDS; lane k of wave 0 adds node k's S records in
// slice order and runs node k's part of the recurrence (the other waves wait at the closing
// barrier).  Protocol as in rlvi_coop.h (sc1 stores / loads, parity slots, tags from the
// workspace base, wall-clock-bounded spins).
// ---------------------------------------------------------------------------------------

template <bool FIRST>
__device__ __forceinline__ void tj_round(TjShared &sh, float fS, float fP, float fQ, float fD,
                                         float fmin_,
                                         gu64 *slots, uint32_t tag, int xstep, int K, int S,
                                         int32_t *status, bool &dead, float rn_l, float shift,
                                         float invN, float tol, float *trace, bool want_nodes,
                                         unsigned long long *dbg = nullptr) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    // wave level in fp32 (one fused v_add_f32_dpp per butterfly step; <= 704 terms per wave, error
    // ~1e-7 relative, below the fp32 rounding of mean(pi) itself), fp64 across waves / workgroups
    const float a = group_allreduce<WAVE>(fS, FAdd());
    const float b = group_allreduce<WAVE>(fP, FAdd());
    const float c = group_allreduce<WAVE>(fQ, FAdd());
    const float d4 = group_allreduce<WAVE>(fD, FAdd());
    float mn = FIRST ? group_allreduce<WAVE>(fmin_, FMin()) : 0.0f;
    if (lane == 0) {
        sh.part[wave][0] = (double)a; sh.part[wave][1] = (double)b; sh.part[wave][2] = (double)c;
        sh.part[wave][3] = (double)d4;
        if (FIRST) sh.pmin[wave] = mn;
    }
    __syncthreads();
#define TJ_RS(i) do { if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && xstep < 4) dbg[200 + xstep * 8 + (i)] = wall_clock64(); } while (0)
    TJ_RS(0);
    constexpr int NQ = FIRST ? 6 : 5;                // granules: S lo, S hi, S', Q, D, (min)
    constexpr int PER = MAX_COOP_WG / WAVE;          // polling waves (one record per lane each)
    auto dbl = [](unsigned long long lo, unsigned long long hi) {
        return __longlong_as_double((long long)(((hi & 0xFFFFFFFFull) << 32) | (lo & 0xFFFFFFFFull)));
    };
    gu64 *buf = slots + (size_t)(xstep & 1) * MAX_COOP_WG * XCHG2_GRANULES;
    const int nwg = K * S;
    if (wave == 0 && !dead) {
        double t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = lane < TJ_NW ? sh.part[lane][q] : 0.0;
            t[q] = group_allreduce<WAVE>(v, FAdd());
        }
        const float tmin = FIRST ? group_allreduce<WAVE>(lane < TJ_NW ? sh.pmin[lane] : __builtin_inff(), FMin())
                                 : 0.0f;
        if (lane < NQ) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(t[0]);
            uint32_t half;
            if (lane == 0) half = (uint32_t)bits;
            else if (lane == 1) half = (uint32_t)(bits >> 32);
            else if (lane == 2) half = __float_as_uint((float)t[1]);
            else if (lane == 3) half = __float_as_uint((float)t[2]);
            else if (lane == 4) half = __float_as_uint((float)t[3]);
            else half = __float_as_uint(tmin);
            __hip_atomic_store(buf + (size_t)blockIdx.x * XCHG2_GRANULES + lane,
                               ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    TJ_RS(1);
    // waves 0..3 sweep 64 records each (one record per lane, all its granules in flight) into LDS
    if (wave < PER && !dead) {
        const int w = wave * WAVE + lane;
        const bool mine = w < nwg;
        gu64 *p = buf + (size_t)(mine ? w : 0) * XCHG2_GRANULES;
        const unsigned long long t0 = wall_clock64();
        bool timeout = false;
        unsigned long long x[8];
        for (unsigned spin = 0;; ++spin) {
            bool ok = true;
            if (mine) {
                // the 64-byte record as four 16-byte agent-scope (sc1) loads in flight together:
                // half the requests of 8-byte loads; each 8-byte granule is self-tagged, so it does
                // not matter that a 16-byte load is only granule-atomic
                vu4 q0, q1, q2, q3;
                asm volatile(
                    "global_load_dwordx4 %0, %4, off sc1\n\t"
                    "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                    "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                    "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                    : "v"((unsigned long long)(uintptr_t)p)
                    : "memory");
                x[0] = ((unsigned long long)q0.y << 32) | q0.x; x[1] = ((unsigned long long)q0.w << 32) | q0.z;
                x[2] = ((unsigned long long)q1.y << 32) | q1.x; x[3] = ((unsigned long long)q1.w << 32) | q1.z;
                x[4] = ((unsigned long long)q2.y << 32) | q2.x; x[5] = ((unsigned long long)q2.w << 32) | q2.z;
                x[6] = ((unsigned long long)q3.y << 32) | q3.x; x[7] = ((unsigned long long)q3.w << 32) | q3.z;
#pragma unroll
                for (int q = 0; q < NQ; ++q) ok = ok && (uint32_t)(x[q] >> 32) == tag;
            }
            if (__all(ok)) break;
            if ((spin & 63u) == 63u && wall_clock64() - t0 > SPIN_BOUND_TICKS) {
                timeout = true;
                break;
            }
        }
        if (timeout) {
            if (lane == 0) { atomicOr(status, RLVI_ST_TIMEOUT); sh.out.dead = 1; }
        } else if (mine) {
            sh.rec[w][0] = dbl(x[0], x[1]);
#pragma unroll
            for (int q = 1; q < NQ - 1; ++q) sh.rec[w][q] = (double)__uint_as_float((uint32_t)x[q + 1]);
        }
    }
    TJ_RS(2);
    __syncthreads();
    TJ_RS(3);
    dead = dead || sh.out.dead != 0;
    if (wave == 0) {
        double tS = 0.0, tP = 0.0, tQ = 0.0, tD = 0.0;
        float gmin = __builtin_inff();
        if (!dead && lane < K) {
            for (int ss = 0; ss < S; ++ss) {              // fixed order over the slices
                const int w = lane * S + ss;
                tS += sh.rec[w][0];
                tP += sh.rec[w][1];
                tQ += sh.rec[w][2];
                tD += sh.rec[w][3];
                if constexpr (FIRST) gmin = fminf(gmin, (float)sh.rec[w][4]);
            }
        }
        TJ_RS(4);
        tj_chain<FIRST, true>(sh.out, K, K, tS, tP, tQ, tD, gmin, dead, rn_l, shift, invN, tol, trace,
                        want_nodes, xstep, dbg);
        TJ_RS(6);
    }
    __syncthreads();
    dead = sh.out.dead != 0;
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
    if (threadIdx.x == 0) sh.out.dead = 0;
    char *wsb = static_cast<char *>(ws);
    WsHeader *hdr = reinterpret_cast<WsHeader *>(wsb);
    gu64 *slots = (gu64 *)(reinterpret_cast<unsigned long long *>(wsb + WS_XCHG2_OFF));
    TrajState *state = reinterpret_cast<TrajState *>(wsb + WS_TRAJ_OFF);
    uint32_t tag = __hip_atomic_load((gu32 *)&hdr->epoch_base, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) + 1u;
    int xstep = 0;
    bool dead = false;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int k = (int)blockIdx.x / S, s = (int)blockIdx.x - k * S;
    const int64_t L = (N + S - 1) / S;
    const int64_t lo = (int64_t)s * L;
    const int64_t hi = lo + L < N ? lo + L : N;
    constexpr int EE = E > 0 ? E : 1;

    // ---- warm-start state (read early: its latency hides behind the slice loads)
    const bool warm = state->n == (long long)N && state->k == K;
    const float shift = warm ? state->shift : 0.0f;     // guess of min(l); NLLs are >= 0
    // node guesses in r-space: last call's trajectory, else geometric; lane j of every wave holds
    // node j (wave 0 needs them all, the workgroup needs r_k and r_{k-1})
    float rn_l = lane < K ? (warm ? state->nodes[lane] : 19.0f * exp2f(-(float)lane)) : 1.0f;
    if (lane == 0) rn_l = (float)(0.95 / (1.0 - 0.95));

    // ---- slice -> registers (raw residuals), local min
    float e[EE];
    int cnt = 0;
    float mn = __builtin_inff();
    if (E > 0) {
#pragma unroll
        for (int j = 0; j < EE; ++j) {
            const int64_t i = lo + tid + (int64_t)j * TJ_BLOCK;
            const bool ok = i < hi;
            e[j] = ok ? res[i] : __builtin_inff();
            if (ok) { cnt = j + 1; mn = fminf(mn, e[j]); }
        }
    } else {
        for (int64_t i = lo + tid; i < hi; i += TJ_BLOCK) mn = fminf(mn, res[i]);
    }
    TJ_STAMP();   // slice loaded

    const float invN = 1.0f / (float)N;
    float r_mine = __shfl(rn_l, k, WAVE);
    float r_prev = k > 0 ? __shfl(rn_l, k - 1, WAVE) : 0.0f;
    float cshift = shift;        // shift the e values are currently computed with
    bool e_ready = false;        // e[] holds exp(-(l - true min)) (after round 0)
    int it = K;
    float r_fin = r_mine;
    float gmin = 0.0f;
    bool accepted = false;
    for (int round = 0; round <= K + 1; ++round) {
        // ---- sums of this workgroup's node over its slice
        float fS = 0.0f, fP = 0.0f, fQ = 0.0f, fD = 0.0f;
        auto body = [&](float ev, float wv) {
            const float t = r_mine * ev;
            const float inv = __builtin_amdgcn_rcpf(1.0f + t);
            const float f = t * inv;
            float fp;
            if (k == 0) fp = wv;                           // caller's pi: D_0 (:33, first pass)
            else { const float tp = r_prev * ev; fp = tp * __builtin_amdgcn_rcpf(1.0f + tp); }
            const float d = f - fp;
            fS += f;
            const float xq = ev * inv, yq = xq * inv;
            fP += yq;                                      // e/(1+re)^2
            fQ = fmaf(xq, yq, fQ);                         // e^2/(1+re)^3
            fD += d * d;
        };
        if (E > 0) {
#pragma unroll
            for (int j = 0; j < EE; ++j) {
                if (j < cnt) {
                    const float ev = e_ready ? e[j] : expf(-(e[j] - cshift));
                    const float wv = k == 0 ? wts[lo + tid + (int64_t)j * TJ_BLOCK] : 0.0f;
                    body(ev, wv);
                }
            }
        } else {
            // streaming form: four independent loads in flight per thread
            for (int64_t i = lo + tid; i < hi; i += 4 * TJ_BLOCK) {
                float lv[4], wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t ii = i + (int64_t)u * TJ_BLOCK;
                    lv[u] = ii < hi ? res[ii] : __builtin_inff();
                    wv[u] = (k == 0 && ii < hi) ? wts[ii] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + (int64_t)u * TJ_BLOCK < hi) body(expf(-(lv[u] - cshift)), wv[u]);
            }
        }
        TJ_STAMP();   // sums done
        if (round == 0)
            tj_round<true>(sh, fS, fP, fQ, fD, mn, slots, tag, xstep, K, S, &hdr->status, dead, rn_l,
                           shift, invN, tol, trace, true, dbg);
        else
            tj_round<false>(sh, fS, fP, fQ, fD, 0.0f, slots, tag, xstep, K, S, &hdr->status, dead, rn_l,
                            shift, invN, tol, trace, true, dbg);
        ++tag; ++xstep;
        TJ_STAMP();   // exchange + recurrence done
        it = sh.out.res_it;
        r_fin = sh.out.res_rfin;
        const float delta = sh.out.res_delta;
        if (dbg != nullptr && blockIdx.x == 0 && tid == 0 && round < 24 && round == 0) { dbg[100] = __float_as_uint(sh.out.res_min); dbg[101] = __float_as_uint(sh.out.res_rfin); }
        if (dbg != nullptr && blockIdx.x == 0 && tid == 0 && round < 24)
            dbg[64 + round] = ((unsigned long long)it << 32) | __float_as_uint(delta);
        rn_l = lane < K ? sh.out.nodes[lane] : 1.0f;
        if (lane == 0) rn_l = (float)(0.95 / (1.0 - 0.95));
        r_mine = __shfl(rn_l, k, WAVE);
        r_prev = k > 0 ? __shfl(rn_l, k - 1, WAVE) : 0.0f;
        if (round == 0) {
            // the true minimum is known now: residuals.sub_(min) (:27), e = exp(-residuals) (:28)
            gmin = sh.out.res_min;
            cshift = gmin;
            if (E > 0) {
#pragma unroll
                for (int j = 0; j < EE; ++j) {
                    const float l = e[j] - gmin;
                    if (j < cnt && k == 0) res[lo + tid + (int64_t)j * TJ_BLOCK] = l;
                    e[j] = j < cnt ? expf(-l) : 0.0f;
                }
                e_ready = true;
            }
        }
        __syncthreads();          // sh.out.nodes / sh.out.res_* are rewritten next round
        if (delta <= TJ_ACCEPT) { accepted = true; break; }
        if (dead) break;
    }
    // K+2 rounds always suffice (every round makes one more node exact); anything else is a bug
    // or a non-finite input: report it instead of returning silently wrong posteriors
    if (!accepted && tid == 0) atomicOr(&hdr->status, RLVI_ST_NOCONV);

    // ---- weights = pi / max(pi); max is attained at e = 1 (the min-residual sample) (:38)
    // f_max is evaluated with the very expression used per element and the normalisation is a
    // true division, so the min-residual sample (e = 1) comes out as exactly 1.0, as div_(max) does
    const float tmax = r_fin * 1.0f;
    // (a solve that did not converge -- a non-finite residual -- poisons every weight, as the
    //  reference's min / mean over a vector with a NaN does)
    const float pmax = accepted ? tmax * __builtin_amdgcn_rcpf(1.0f + tmax) : __builtin_nanf("");
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
                    wts[i] = (t * __builtin_amdgcn_rcpf(1.0f + t)) / pmax;
                }
                pm += stepm;
                pm = pm >= K ? pm - K : pm;
            }
        } else {
            for (int64_t i = lo + tid; i < hi; i += TJ_BLOCK) {
                if (pm == k) {
                    const float l = res[i] - gmin;
                    const float t = r_fin * expf(-l);
                    res[i] = l;
                    wts[i] = (t * __builtin_amdgcn_rcpf(1.0f + t)) / pmax;
                }
                pm += stepm;
                pm = pm >= K ? pm - K : pm;
            }
        }
    }
    TJ_STAMP();   // final stores issued
    if (dbg != nullptr && blockIdx.x == 0 && tid == 0) dbg[63] = (unsigned long long)dbgi;
    if (blockIdx.x == 0 && tid < WAVE) {
        if (tid == 0) {
            if (out_iters != nullptr) *out_iters = it;
            state->n = (long long)N;
            state->k = K;
            state->shift = gmin;
            state->it = it;
            __hip_atomic_store((gu32 *)&hdr->epoch_base, tag + 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < K) state->nodes[tid] = rn_l;
    }
}

// Eligibility + geometry.  Returns 1 if launched (rc in *rc), 0 if the caller should use the
// iterative kernel.
int try_launch_estep_traj(float *res, float *wts, int64_t N, float tol, int maxiter,
                          int32_t *out_iters, float *trace, void *ws, hipStream_t st,
                          float *mstep_out, double mstep_scale, int *rc) {
    static const int mode = getenv("RLVI_ESTEP_TRAJ") ? atoi(getenv("RLVI_ESTEP_TRAJ")) : 1;
    // (from 12 288 samples on estep_trajb.hip is tried first and wins; this bound only matters when
    //  that kernel is switched off)
    static const int64_t nmax = getenv("RLVI_ESTEP_TRAJ_NMAX") ? atoll(getenv("RLVI_ESTEP_TRAJ_NMAX")) : 200000;
    if (mode == 0 || maxiter < 1 || maxiter > TJ_MAXK || N < 4096 || N > nmax) return 0;
    const int K = maxiter;
    int S = (MAX_COOP_WG - 1) / K;
    if (S < 1) return 0;
    if (S > TJ_MAXS) S = TJ_MAXS;
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
    else if (L <= (int64_t)TJ_BLOCK * 12) RLVI_TJ(12);
    else if (L <= (int64_t)TJ_BLOCK * 16) RLVI_TJ(16);     // (spills a few registers; still far
                                                           //  ahead of re-streaming the slice)
    else RLVI_TJ(0);
#undef RLVI_TJ
    *rc = (int)hipGetLastError();
    return 1;
}

}  // namespace rlvi
