// The cooperating-workgroup trajectory solve of the E-step (train_rlvi.py:14-38), shared by the
// stand-alone E-step kernel (estep_trajb.hip) and the one-launch in-batch E+M kernel (fused_em.hip).
// The caller owns the slice: it hands over the raw residuals and the caller's pi in registers and gets
// back r = eps/(1-eps) at the fixed point, the global minimum and e = exp(-(l - min)); what it does
// with them (store pi, or go on to the weighted gradient) is its business.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "rlvi_traj.h"

namespace rlvi {

typedef unsigned int tb_vu4 __attribute__((ext_vector_type(4)));

// Workgroup size: 256 threads = one wave per SIMD (the per-wave butterflies are a fixed cost per
// wave and chunk, so fewer, fatter waves win as long as the slice fits the registers); the main loop
// has no memory operations to hide and eight independent nodes of instruction-level parallelism.
#ifndef RLVI_TB_G
#define RLVI_TB_G 256
#endif
// 256 exchanging workgroups (the epoch-end reduction rides on the last one): at the bench size a slice is
// exactly one sample per thread (32.6 us per step against 33.9 with 240).
constexpr int TB_G = RLVI_TB_G;      // exchanging workgroups at most (= exchange slots per node)
constexpr int TB_CHUNK = 8;
constexpr int TB_NV = 8;             // values of a record: {S, P, Q, D, min, R3, R4, -}
constexpr int TB_PER = (TB_G + WAVE - 1) / WAVE;   // polling waves of a stage-A gather

// Node-split sums (slices up to TB_NSPLIT_MAXS samples per workgroup): the slice's e = exp(-(l - min)) and
// the caller's pi are staged through LDS so that EVERY wave sees all samples and the waves share out the
// NODES instead of the samples (see `sums_split` below).
constexpr int TB_NSPLIT_MAXS = 1024;     // node-split up to this many samples per workgroup (16 per lane)
constexpr bool tb_nsplit(int E, int block) { return E * block <= TB_NSPLIT_MAXS; }
constexpr int tb_stage(int E, int block) { return tb_nsplit(E, block) ? E * block : 4; }

// STAGE > 4 <=> node-split: then no wave partials go through LDS at all (records leave from the registers)
template <int TB_NW, int STAGE = 4>
struct TbShared {
    float wp[STAGE > 4 ? 1 : TB_NW][STAGE > 4 ? 1 : TJ_MAXK][8];     // wave partials {S, P, Q, D, R3, R4, P2, -} per node (sample-split sums)
    float pmin[TB_NW];
    double red[TB_PER][TB_NV];
    TjOut out;
    alignas(16) float es[STAGE];     // e of every sample of the slice (node-split sums)
    alignas(16) float qs[STAGE];     // the caller's pi of every sample (D_0)
};

// v[q]: this lane's partial of node q of a chunk.  Returns, in lane l, the wave total of node
// (l >> 3) & 7.  Fixed pairing order: deterministic.
__device__ __forceinline__ float wave_reduce8(const float (&v)[TB_CHUNK]) {
    float u[4], w[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {     // lanes < 32 keep node i, lanes >= 32 node i + 4
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        u[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {     // even rows keep node i (+4), odd rows node i + 2 (+4)
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[i]), __float_as_uint(u[i + 2]), false, false);
        w[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const bool hi8 = (threadIdx.x & 8) != 0;
    const float send = hi8 ? w[0] : w[1];
    const float keep = hi8 ? w[1] : w[0];
    const float x = keep + dpp_x<0x128>(send);            // row_ror:8
    return group_allreduce<8>(x, FAdd());
}

// 48-byte record = six granules {tag32 | payload32}: {S, P, Q, D, min, -}; the first nq (4 or 5)
// must carry `tag`.  Each 8-byte granule is self-tagged, so it does not matter that a 16-byte load
// is only granule-atomic.
__device__ __forceinline__ bool load_rec(gu64 *p, uint32_t tag, int nq, float (&val)[TB_NV]) {
    // nq granules of the record carry this step's tag: 4 {S, P, Q, D}, 5 {.., min} or 7 {.., R3, R4}
    tb_vu4 q0, q1, q2, q3;
    q2.x = 0u; q2.y = tag; q2.z = 0u; q2.w = tag;
    q3.x = 0u; q3.y = tag; q3.z = 0u; q3.w = tag;
    if (nq > 5) {
        asm volatile(
            "global_load_dwordx4 %0, %4, off sc1\n\t"
            "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
            "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
            "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
            : "v"((unsigned long long)(uintptr_t)p)
            : "memory");
        q3.w = tag;                      // (granule 7 is not part of a 7-granule record)
    } else if (nq > 4) {
        asm volatile(
            "global_load_dwordx4 %0, %3, off sc1\n\t"
            "global_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
            "global_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(q0), "=&v"(q1), "=&v"(q2)
            : "v"((unsigned long long)(uintptr_t)p)
            : "memory");
        q2.w = tag;                      // (granule 5 is not part of a 5-granule record)
    } else
        asm volatile(
            "global_load_dwordx4 %0, %2, off sc1\n\t"
            "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(q0), "=&v"(q1)
            : "v"((unsigned long long)(uintptr_t)p)
            : "memory");
    val[0] = __uint_as_float(q0.x); val[1] = __uint_as_float(q0.z);
    val[2] = __uint_as_float(q1.x); val[3] = __uint_as_float(q1.z);
    val[4] = __uint_as_float(q2.x); val[5] = __uint_as_float(q2.z);
    val[6] = __uint_as_float(q3.x); val[7] = __uint_as_float(q3.z);
    return q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag && q2.y == tag && q2.w == tag &&
           q3.y == tag && q3.w == tag;
}

__device__ __forceinline__ void store_rec(gu64 *p, uint32_t tag, int nq, const float (&val)[TB_NV]) {
#pragma unroll
    for (int q = 0; q < TB_NV; ++q)
        if (q < nq)
            __hip_atomic_store(p + q, ((unsigned long long)tag << 32) | __float_as_uint(val[q]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// warm-start state of the previous call (read it ahead of the slice loads: its latency hides there)
struct TbWarm {
    bool warm;
    float shift;      // guess of min(l); NLLs are >= 0
    float rn_l;       // lane k: guessed node k
    int it;
};
__device__ __forceinline__ TbWarm tb_warm(void *ws, int64_t N, int K, bool sharded = false, bool cold = false) {
    const TrajState *state = reinterpret_cast<const TrajState *>(static_cast<char *>(ws) +
                                                                 (sharded ? WS_PEER_STATE_OFF : WS_TRAJ_OFF));
    const int lane = threadIdx.x & (WAVE - 1);
    TbWarm w;
    w.warm = !cold && state->n == (long long)N && state->k == K;
    w.shift = w.warm ? state->shift : 0.0f;
    w.rn_l = lane < K ? (w.warm ? state->nodes[lane] : 19.0f * exp2f(-(float)lane)) : 1.0f;
    if (lane == 0) w.rn_l = (float)(0.95 / (1.0 - 0.95));
    w.it = w.warm ? state->it : 0;
    return w;
}

struct TbSolved {
    float r_fin;      // eps/(1-eps) at the accepted fixed point
    float gmin;       // min(l) over all samples
    int it;           // iterations the reference's loop would have taken
    bool accepted, dead;
    uint32_t tag_free;   // an exchange tag no round of this launch (and no later launch) uses
};

// pi / max(pi) of a sample with e = exp(-(l - min)); the maximum is attained at e = 1 (:38): the same
// expression per element and for the maximum and a true division, so that sample is exactly 1.0.
// (A solve that did not converge -- a non-finite residual -- poisons every weight, as the
//  reference's min / mean over a vector with a NaN does.)
__device__ __forceinline__ float tb_pmax(const TbSolved &s) {
    const float tmax = s.r_fin * 1.0f;
    return s.accepted ? tmax * __builtin_amdgcn_rcpf(1.0f + tmax) : __builtin_nanf("");
}
__device__ __forceinline__ float tb_weight(const TbSolved &s, float pmax, float e) {
    const float t = s.r_fin * e;
    return (t * __builtin_amdgcn_rcpf(1.0f + t)) / pmax;
}

// All threads of the workgroup call (the barriers inside are workgroup barriers); `active` threads
// (threadIdx.x < NGRP * TB_BLOCK) hold slice element (tid % TB_BLOCK) + j TB_BLOCK in l[j] / q0[j].
// NGRP > 1: the caller has spare waves; thread groups of TB_BLOCK hold copies of the slice and share
// out the node chunks of the per-node sums (the exchange and the recurrence stay with group 0).
// Workgroup b of G.
template <int E, int TB_BLOCK, int NGRP = 1>
__device__ __forceinline__ TbSolved trajb_solve(
    TbShared<TB_BLOCK / WAVE, tb_stage(E, TB_BLOCK)> &sh, const TbWarm &wm, const float (&l)[E], const float (&q0)[E],
    float (&ev)[E], const bool active, const int b, const int G, const int64_t N, const float tol,
    const int K, int32_t *__restrict__ out_iters, float *__restrict__ trace, void *ws,
    unsigned long long *__restrict__ dbg, PeerTable *__restrict__ pt = nullptr, const bool verify = false,
    float *__restrict__ red_out = nullptr, const double red_scale = 1.0) {
    int dbgi = 0;
#define TB_STAMP() do { if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && threadIdx.x == 0 && dbgi < 60) dbg[dbgi++] = wall_clock64(); } while (0)
    TB_STAMP();
    constexpr int TB_NW = TB_BLOCK / WAVE;
    static_assert(TB_NW >= TB_PER, "the stage-A gather needs four waves");
    if (threadIdx.x == 0) sh.out.dead = 0;
    char *wsb = static_cast<char *>(ws);
    WsHeader *hdr = reinterpret_cast<WsHeader *>(wsb);
    gu64 *bufA = (gu64 *)(reinterpret_cast<unsigned long long *>(wsb + WS_XCHG3A_OFF));
    gu64 *bufB = (gu64 *)(reinterpret_cast<unsigned long long *>(wsb + WS_XCHG3B_OFF));
    TrajState *state = reinterpret_cast<TrajState *>(wsb + (pt != nullptr ? WS_PEER_STATE_OFF : WS_TRAJ_OFF));
    uint32_t tag = __hip_atomic_load((gu32 *)&hdr->epoch_base, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // (sharded over several GPUs: a peer may legitimately be late -- another process, another stream --,
    //  and every wait downstream of the cross-rank hop inherits its lateness: 100 x the bound)
    const unsigned long long spin_ticks = spin_bound(hdr) * (pt != nullptr ? 100ull : 1ull);
    int xstep = 0;
    bool dead = false;
    // sharded over several GPUs (pt != nullptr): N is the population over ALL ranks; the reducer of a
    // node pushes this rank's total into every rank's inbox and adds up what the others pushed.
    // ptag numbers the rounds of all sharded solves of this group: the same on every rank.
    uint32_t ptag = pt != nullptr ? pt->dtag + 1u : 0u;
    const int pworld = pt != nullptr ? pt->world : 1;
    const int prank = pt != nullptr ? pt->rank : 0;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid / WAVE;
    const int swave = wave % (TB_BLOCK / WAVE);      // wave inside its sampling group
    const int sgrp = wave / (TB_BLOCK / WAVE);       // sampling group
    // ---- warm-start state (the caller read it ahead of its slice loads)
    const float shift = wm.shift;
    float rn_l = wm.rn_l;
    auto round8 = [K](int v) { v = (v + TB_CHUNK - 1) / TB_CHUNK * TB_CHUNK; return v < K ? v : K; };
    int Ke = __builtin_amdgcn_readfirstlane(wm.warm ? round8(wm.it + 2) : K);            // evaluated nodes (uniform)

    // ---- the slice is in registers (pads: l = +inf, q0 = 0): e' and the local minimum
    float mn = __builtin_inff();
#pragma unroll
    for (int j = 0; j < E; ++j) mn = fminf(mn, l[j]);
#pragma unroll
    for (int j = 0; j < E; ++j) ev[j] = expf(-(l[j] - shift));      // pads: exp(-inf) = 0
    mn = group_allreduce<WAVE>(mn, FMin());
    if (lane == 0 && active && sgrp == 0) sh.pmin[wave] = mn;
    // Node-split sums: with few samples per thread the per-node sums are not arithmetic but the
    // transposing butterflies -- a fixed cost per WAVE and node chunk (19 cross-lane operations per
    // quantity and 8 nodes), paid by every wave for all nodes.  So the slice goes through LDS once, every
    // lane takes SPL = 4 E consecutive samples of the WHOLE slice, and the waves share out the nodes: wave w
    // takes nodes [w npw, (w + 1) npw), npw = ceil(Ke / waves) -- a quarter of the butterflies per wave (an
    // eighth with the in-batch kernel's eight waves), the same arithmetic per lane, sample pairs in packed
    // fp32 even at one sample per thread, and a node's totals come out of ONE wave (no cross-wave sum).
    constexpr bool NSPLIT = tb_nsplit(E, TB_BLOCK);
    constexpr int SPL = NSPLIT ? E * TB_BLOCK / WAVE : 4;      // samples per lane
    constexpr int NWS = NGRP * (TB_BLOCK / WAVE);              // waves that share out the nodes
    auto stage_slice = [&](bool with_q) {
        if constexpr (NSPLIT) {
            if (active && sgrp == 0) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    sh.es[tid + j * TB_BLOCK] = ev[j];
                    if (with_q) sh.qs[tid + j * TB_BLOCK] = q0[j];
                }
            }
            __syncthreads();
        }
    };
    stage_slice(true);
#if RLVI_STAMPS
    if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { dbg[983] = __builtin_amdgcn_s_memtime(); dbg[985] = wall_clock64(); }
#endif
    // (node-split: the workgroup's minimum is known to every wave from here on -- each wave publishes the
    //  records of its own nodes itself, the minimum included)
    float wmin_all = __builtin_inff();
    if constexpr (NSPLIT) {
#pragma unroll
        for (int w = 0; w < TB_NW; ++w) wmin_all = fminf(wmin_all, sh.pmin[w]);
    }
    TB_STAMP();   // slice loaded

    const float invN = 1.0f / (float)N;
    int it = K;
    float r_fin = rn_l;
    float gmin = 0.0f;
    bool accepted = false;
    const int max_rounds = 2 * K + 2;
    for (int round = 0; round < max_rounds; ++round) {
        // ---- per-node sums over this workgroup's slice, eight nodes at a time
        float fprev[E];
#pragma unroll
        for (int j = 0; j < E; ++j) fprev[j] = q0[j];      // "node -1" = the caller's pi (D_0)
        const int nchunks = (Ke + TB_CHUNK - 1) / TB_CHUNK;
        // HI: the first round also takes R3 = sum e^3/(1+re)^4 and R4 = sum e^4/(1+re)^5, the third- and
        // fourth-order terms of S around the node: with them the corrected nodes are good enough (and
        // provably so) for tj_chain to accept without a verification round
        constexpr bool HI_OK = TB_BLOCK == 256;
        // (verify -- RLVI_TJ_VERIFY=1 -- forces the verification round: no fourth-order first round, no
        //  acceptance on estimated step errors; the tests hold the two paths against each other)
        // (... and none without a guess: a cold start's first round goes to the global model whatever its sums
        //  carry -- the two extra accumulators and granules would be 0.8 us for nothing)
        const bool hi_round = HI_OK && round == 0 && trace == nullptr && !verify && wm.warm;
        gu64 *A = bufA + (size_t)(xstep & 1) * TJ_MAXK * MAX_COOP_WG * XCHG3_GRANULES;
        // (every replica on its own 3-KiB stretch: 256 pollers on one 768-byte stretch serialise at the
        //  memory side)
        gu64 *B = bufB + (size_t)(xstep & 1) * XCHG3B_REPLICAS * TJ_MAXK * XCHG3_GRANULES;
        const int nq = hi_round ? 7 : (round == 0 ? 5 : 4);      // granules of a record that carry this step's tag
        auto sums = [&](auto hi_tag) {
            constexpr bool HI = decltype(hi_tag)::value;
#pragma unroll 1
            for (int c = NGRP > 1 ? sgrp : 0; c < nchunks; c += NGRP) {
                float aI[TB_CHUNK], aP[TB_CHUNK], aQ[TB_CHUNK], aD[TB_CHUNK];
                float a3[TB_CHUNK], a4[TB_CHUNK];
                if (NGRP > 1 && c > 0) {
                    // the chunk before this one was another group's: pi at its last node again (same
                    // operations as below, same bits)
                    const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rn_l), c * TB_CHUNK - 1));
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        const float t = r * ev[j];
                        fprev[j] = t * __builtin_amdgcn_rcpf(1.0f + t);
                    }
                }
#pragma unroll
                for (int q = 0; q < TB_CHUNK; ++q) {
                    const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rn_l), c * TB_CHUNK + q));
                    // sample pairs in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two
                    // samples per lane and instruction); only the reciprocal is per sample
                    f32x2_t pS = {0.0f, 0.0f}, pP = {0.0f, 0.0f}, pQ = {0.0f, 0.0f}, pD = {0.0f, 0.0f};
                    f32x2_t p3 = {0.0f, 0.0f}, p4 = {0.0f, 0.0f};
                    const f32x2_t r2 = {r, r}, one2 = {1.0f, 1.0f};
#pragma unroll
                    for (int j = 0; j + 1 < E; j += 2) {
                        const f32x2_t e2 = {ev[j], ev[j + 1]};
                        const f32x2_t t = r2 * e2, t1 = t + one2;
                        const f32x2_t inv = {__builtin_amdgcn_rcpf(t1.x), __builtin_amdgcn_rcpf(t1.y)};
                        const f32x2_t f = t * inv;                                       // pi (:30)
                        pS += f;
                        const f32x2_t x = e2 * inv, y = x * inv;
                        pP += y;                                                         // e/(1+re)^2
                        const f32x2_t xy = x * y;
                        pQ += xy;                                                        // e^2/(1+re)^3
                        if (HI) {
                            const f32x2_t xyx = xy * x;
                            p3 += xyx;                                                   // e^3/(1+re)^4
                            p4 = __builtin_elementwise_fma(xyx, x, p4);                  // e^4/(1+re)^5
                        }
                        const f32x2_t d = f - (f32x2_t){fprev[j], fprev[j + 1]};          // pi_k - pi_{k-1}
                        pD = __builtin_elementwise_fma(d, d, pD);
                        fprev[j] = f.x; fprev[j + 1] = f.y;
                    }
                    float sI = pS.x + pS.y, sP = pP.x + pP.y, sQ = pQ.x + pQ.y, sD = pD.x + pD.y;
                    float s3 = p3.x + p3.y, s4 = p4.x + p4.y;
                    if constexpr ((E & 1) != 0) {
                        constexpr int j = E - 1;
                        const float t = r * ev[j];
                        const float inv = __builtin_amdgcn_rcpf(1.0f + t);
                        const float f = t * inv;
                        sI += f;
                        const float x = ev[j] * inv, y = x * inv;
                        sP += y;
                        const float xy = x * y;
                        sQ += xy;
                        if (HI) {
                            const float xyx = xy * x;
                            s3 += xyx;
                            s4 = fmaf(xyx, x, s4);
                        }
                        const float d = f - fprev[j];
                        sD = fmaf(d, d, sD);
                        fprev[j] = f;
                    }
                    aI[q] = sI; aP[q] = sP; aQ[q] = sQ; aD[q] = sD;
                    a3[q] = s3; a4[q] = s4;
                }
                const float tI = wave_reduce8(aI);
                const float tP = wave_reduce8(aP);
                const float tQ = wave_reduce8(aQ);
                const float tD = wave_reduce8(aD);
                float t3 = 0.0f, t4 = 0.0f;
                if (HI) { t3 = wave_reduce8(a3); t4 = wave_reduce8(a4); }
                if ((lane & 7) == 0) {
                    float *dst = sh.wp[swave][c * TB_CHUNK + (lane >> 3)];
                    *reinterpret_cast<float4 *>(dst) = make_float4(tI, tP, tQ, tD);
                    if (HI) *reinterpret_cast<float2 *>(dst + 4) = make_float2(t3, t4);
                }
            }
        };
        // (only the 256-thread geometry, i.e. slices up to 8192 samples: the fat 512-thread forms have
        //  no registers to spare for three more accumulator sets)
        auto sums_split = [&](auto hi_tag) {
            // (no implicit contraction in here: which a * b + c the compiler fuses may differ between the
            //  kernels this is inlined into and between the two places that evaluate pi at a node -- the
            //  stand-alone E-step and the in-batch kernel must produce the same bits from the same slice)
#pragma clang fp contract(off)
            constexpr bool HI = decltype(hi_tag)::value;
            const int npw = (Ke + NWS - 1) / NWS;
            // (wave-uniform, and told so: the node loop's bounds and the lane selects below stay scalar)
            const int n0 = __builtin_amdgcn_readfirstlane(wave * npw);   // this wave's nodes [n0, n1)
            const int n1 = __builtin_amdgcn_readfirstlane(n0 + npw < Ke ? n0 + npw : Ke);
            if (n0 >= n1) return;
            float e4[SPL], fpv[SPL];
#pragma unroll
            for (int s4 = 0; s4 < SPL; s4 += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(&sh.es[lane * SPL + s4]);
                e4[s4] = v.x; e4[s4 + 1] = v.y; e4[s4 + 2] = v.z; e4[s4 + 3] = v.w;
            }
            if (n0 == 0) {                                     // "node -1" = the caller's pi (D_0)
#pragma unroll
                for (int s4 = 0; s4 < SPL; s4 += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(&sh.qs[lane * SPL + s4]);
                    fpv[s4] = v.x; fpv[s4 + 1] = v.y; fpv[s4 + 2] = v.z; fpv[s4 + 3] = v.w;
                }
            } else {                                           // pi at the node before this wave's first one
                const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rn_l), n0 - 1));
                const f32x2_t r2 = {r, r}, one2 = {1.0f, 1.0f};
#pragma unroll
                for (int s = 0; s < SPL; s += 2) {            // (the very operations of the loop below)
                    const f32x2_t e2 = {e4[s], e4[s + 1]};
                    const f32x2_t t = r2 * e2, t1 = t + one2;
                    const f32x2_t inv = {__builtin_amdgcn_rcpf(t1.x), __builtin_amdgcn_rcpf(t1.y)};
                    const f32x2_t f = t * inv;
                    fpv[s] = f.x; fpv[s + 1] = f.y;
                }
            }
#pragma unroll 1
            for (int c0 = n0; c0 < n1; c0 += TB_CHUNK) {
                float aI[TB_CHUNK], aP[TB_CHUNK], aQ[TB_CHUNK], aD[TB_CHUNK];
                float a3[TB_CHUNK], a4[TB_CHUNK];
#pragma unroll
                for (int q = 0; q < TB_CHUNK; ++q) {
                    aI[q] = 0.0f; aP[q] = 0.0f; aQ[q] = 0.0f; aD[q] = 0.0f; a3[q] = 0.0f; a4[q] = 0.0f;
                    if (c0 + q < n1) {                         // (wave-uniform)
                        const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rn_l), c0 + q));
                        f32x2_t pS = {0.0f, 0.0f}, pP = {0.0f, 0.0f}, pQ = {0.0f, 0.0f}, pD = {0.0f, 0.0f};
                        f32x2_t p3 = {0.0f, 0.0f}, p4 = {0.0f, 0.0f};
                        const f32x2_t r2 = {r, r}, one2 = {1.0f, 1.0f};
#pragma unroll
                        for (int s = 0; s < SPL; s += 2) {
                            const f32x2_t e2 = {e4[s], e4[s + 1]};
                            const f32x2_t t = r2 * e2, t1 = t + one2;
                            const f32x2_t inv = {__builtin_amdgcn_rcpf(t1.x), __builtin_amdgcn_rcpf(t1.y)};
                            const f32x2_t f = t * inv;                                       // pi (:30)
                            pS += f;
                            const f32x2_t x = e2 * inv, y = x * inv;
                            pP += y;                                                         // e/(1+re)^2
                            const f32x2_t xy = x * y;
                            pQ += xy;                                                        // e^2/(1+re)^3
                            if (HI) {
                                const f32x2_t xyx = xy * x;
                                p3 += xyx;                                                   // e^3/(1+re)^4
                                p4 = __builtin_elementwise_fma(xyx, x, p4);                  // e^4/(1+re)^5
                            }
                            const f32x2_t d = f - (f32x2_t){fpv[s], fpv[s + 1]};              // pi_k - pi_{k-1}
                            pD = __builtin_elementwise_fma(d, d, pD);
                            fpv[s] = f.x; fpv[s + 1] = f.y;
                        }
                        aI[q] = pS.x + pS.y; aP[q] = pP.x + pP.y; aQ[q] = pQ.x + pQ.y; aD[q] = pD.x + pD.y;
                        a3[q] = p3.x + p3.y; a4[q] = p4.x + p4.y;
                    }
                }
                if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && threadIdx.x == 0 && round == 0) { dbg[980] = wall_clock64(); dbg[984] = __builtin_amdgcn_s_memtime(); }
                const float tI = wave_reduce8(aI);
                const float tP = wave_reduce8(aP);
                const float tQ = wave_reduce8(aQ);
                const float tD = wave_reduce8(aD);
                float t3 = 0.0f, t4 = 0.0f;
                if (HI) { t3 = wave_reduce8(a3); t4 = wave_reduce8(a4); }
                if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && threadIdx.x == 0 && round == 0) dbg[981] = wall_clock64();
                // stage A straight from the registers: after the butterflies all eight lanes of a group hold
                // their node's totals, so lane 8 s + j stores granule j of node c0 + s -- one write-through
                // store per lane, no LDS, no workgroup barrier, and a wave's records leave as soon as THAT
                // wave is through with its nodes
                {
                    const int sl = lane >> 3, j = lane & 7;
                    const int node = c0 + sl;
                    if (node < n1 && j < nq && !dead) {
                        const float v = j == 0 ? tI : j == 1 ? tP : j == 2 ? tQ : j == 3 ? tD : j == 4 ? wmin_all
                                        : j == 5 ? t3 : t4;
                        __hip_atomic_store(A + ((size_t)node * MAX_COOP_WG + b) * XCHG3_GRANULES + j,
                                           ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        };
        if (active) {     // (threads past TB_BLOCK, if the caller has any, only follow the barriers)
            if constexpr (NSPLIT) {
                if constexpr (HI_OK) {
                    if (hi_round) sums_split(std::true_type{}); else sums_split(std::false_type{});
                } else {
                    sums_split(std::false_type{});
                }
            } else if constexpr (HI_OK) {
                if (hi_round) sums(std::true_type{}); else sums(std::false_type{});
            } else {
                sums(std::false_type{});
            }
        }
        TB_STAMP();   // sums done
        // the epoch end's reduction of the M-step records (train_rlvi.py:86-87,:105), by the last workgroup,
        // whose records are out and which has nothing to do until the totals arrive (it reduces no node
        // unless G == Ke == 64): reads, sums and clears the 1024 x 4 doubles, writes the four scalars
        if (red_out != nullptr && round == 0 && b == G - 1)
            reduce_partials(reinterpret_cast<double *>(wsb + WS_PART_OFF), MSTEP_MAX_BLOCKS, red_scale, red_out, true,
                            NGRP * TB_BLOCK);
        if constexpr (!NSPLIT) __syncthreads();
        // ---- stage A: this workgroup's record of every evaluated node
        // (waves 0..3 each combine the wave partials and store ONE granule per lane -- S, S', Q, D;
        //  a lane's write-through stores go out one after the other)
        if constexpr (!NSPLIT) {
        if (wave < 4 && !dead && lane < Ke) {
            double dq = 0.0;
#pragma unroll
            for (int w = 0; w < TB_NW; ++w) dq += (double)sh.wp[w][lane][wave];
            gu64 *rec = A + ((size_t)lane * MAX_COOP_WG + b) * XCHG3_GRANULES;
            __hip_atomic_store(rec + wave, ((unsigned long long)tag << 32) | __float_as_uint((float)dq),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // second store of a lane: wave 3 the minimum (granule 4), waves 0 and 1 R3 and R4 (granules 5, 6)
            if (wave == 3 && nq > 4) {
                float wmin = sh.pmin[0];
#pragma unroll
                for (int w = 1; w < TB_NW; ++w) wmin = fminf(wmin, sh.pmin[w]);
                __hip_atomic_store(rec + 4, ((unsigned long long)tag << 32) | __float_as_uint(wmin),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((RLVI_STAMPS && dbg != nullptr) && b == 0 && round == 0 && lane == 0) dbg[898] = __float_as_uint(wmin);
            }
            if (wave < 2 && nq > 5) {
                float hq = 0.0f;
#pragma unroll
                for (int w = 0; w < TB_NW; ++w) hq += sh.wp[w][lane][4 + wave];
                __hip_atomic_store(rec + 5 + wave, ((unsigned long long)tag << 32) | __float_as_uint(hq),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        }
        TB_STAMP();   // stage A stored
        // ---- stage B: workgroup k < Ke adds node k's records and publishes the total
        if (b < Ke) {
            if (wave < TB_PER && !dead) {
                const int w = wave * WAVE + lane;
                const bool mine = w < G;
                gu64 *p = A + ((size_t)b * MAX_COOP_WG + (mine ? w : 0)) * XCHG3_GRANULES;
                const unsigned long long t0 = wall_clock64();
                bool timeout = false;
                float val[TB_NV] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                for (unsigned spin = 0;; ++spin) {
                    const bool ok = load_rec(p, tag, nq, val) || !mine;
                    if (__all(ok)) break;
                    if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) {
                        timeout = true;
                        break;
                    }
                }
                if (timeout) {
                    if (lane == 0) { atomicOr(&hdr->status, RLVI_ST_TIMEOUT); sh.out.dead = 1; }
                }
                TB_STAMP();   // stage A gathered
                const double vS = group_allreduce<WAVE>(mine ? (double)val[0] : 0.0, FAdd());
                // (S steers the trajectory: fp64; the slopes and the step error are fine in fp32 -- one
                //  fused v_add_f32_dpp per butterfly step instead of two moves and an fp64 add)
                const double vP = (double)group_allreduce<WAVE>(mine ? val[1] : 0.0f, FAdd());
                const double vQ = (double)group_allreduce<WAVE>(mine ? val[2] : 0.0f, FAdd());
                const double vD = (double)group_allreduce<WAVE>(mine ? val[3] : 0.0f, FAdd());
                const float vM = group_allreduce<WAVE>((mine && nq > 4) ? val[4] : __builtin_inff(), FMin());
                double v3 = 0.0, v4 = 0.0;
                if (nq > 5) {
                    v3 = (double)group_allreduce<WAVE>(mine ? val[5] : 0.0f, FAdd());
                    v4 = (double)group_allreduce<WAVE>(mine ? val[6] : 0.0f, FAdd());
                }
                if (lane == 0) {
                    sh.red[wave][0] = vS; sh.red[wave][1] = vP; sh.red[wave][2] = vQ;
                    sh.red[wave][3] = vD; sh.red[wave][4] = (double)vM;
                    sh.red[wave][5] = v3; sh.red[wave][6] = v4;
                }
            }
            __syncthreads();
            if (wave == 0 && !dead && sh.out.dead == 0) {
                double tS = 0.0, tP = 0.0, tQ = 0.0, tD = 0.0, tM = (double)__builtin_inff();
                double t3 = 0.0, t4 = 0.0;
#pragma unroll
                for (int w = 0; w < TB_PER; ++w) {            // fixed order
                    tS += sh.red[w][0]; tP += sh.red[w][1]; tQ += sh.red[w][2]; tD += sh.red[w][3];
                    tM = sh.red[w][4] < tM ? sh.red[w][4] : tM;
                    t3 += sh.red[w][5]; t4 += sh.red[w][6];
                }
                // lane l stores granule l & 7 of replica l >> 3: one store per lane
                static_assert(XCHG3B_REPLICAS * 8 == WAVE && TB_NV <= 8, "one granule of one replica per lane");
                const int gq = lane & 7;
                float val = gq == 0 ? (float)tS : gq == 1 ? (float)tP : gq == 2 ? (float)tQ
                            : gq == 3 ? (float)tD : gq == 4 ? (float)tM : gq == 5 ? (float)t3
                            : (float)t4;
                bool xdead = false;
                if (pt != nullptr) {
                    // lane = (rank r = lane >> 3, granule gq): one system-scope store into rank r's inbox
                    // slot [round parity][node b][this rank], then this rank's own slots [..][rank r]
                    // until every rank's granules carry this round's tag; totals in rank order (the same
                    // tree on every rank: identical bits everywhere)
                    const int r = lane >> 3;
                    const bool mine = r < pworld && gq < nq;
                    const size_t slot = ((size_t)(ptag & 1u) * TJ_MAXK + b) * MAX_PEERS;
                    if (mine) {
                        gu64 *dst = (gu64 *)(uintptr_t)pt->inbox[r] + (slot + prank) * 8 + gq;
                        __hip_atomic_store(dst, ((unsigned long long)ptag << 32) | __float_as_uint(val),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    gu64 *src = (gu64 *)(uintptr_t)pt->inbox[prank] + (slot + (mine ? r : prank)) * 8 + (mine ? gq : 0);
                    const unsigned long long t0 = wall_clock64();
                    unsigned long long got = 0ull;
                    bool timeout = false;
                    for (unsigned spin = 0;; ++spin) {
                        got = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const bool ok = !mine || (uint32_t)(got >> 32) == ptag;
                        if (__all(ok)) break;
                        if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
                    }
                    if (timeout) {
                        if (lane == 0) { atomicOr(&hdr->status, RLVI_ST_TIMEOUT); sh.out.dead = 1; }
                        xdead = true;
                    }
                    const float pv = __uint_as_float((uint32_t)got);
                    double acc = mine ? (double)pv : (gq == 4 ? (double)__builtin_inff() : 0.0);
#pragma unroll
                    for (int m = 8; m < WAVE; m <<= 1) {
                        const double o = __shfl_xor(acc, m, WAVE);
                        acc = gq == 4 ? (o < acc ? o : acc) : acc + o;
                    }
                    val = (float)acc;
                }
                if (gq < nq && !xdead)
                    __hip_atomic_store(B + ((size_t)(lane >> 3) * TJ_MAXK + b) * XCHG3_GRANULES + gq,
                                       ((unsigned long long)tag << 32) | __float_as_uint(val),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        TB_STAMP();   // published
        // ---- wave 0: the Ke totals (lane k = node k), then the recurrence
        if (wave == 0) {
            dead = dead || sh.out.dead != 0;
            float val[TB_NV] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (!dead) {
                const bool mine = lane < Ke;
                gu64 *p = B + ((size_t)(b & (XCHG3B_REPLICAS - 1)) * TJ_MAXK + (mine ? lane : 0)) * XCHG3_GRANULES;
                const unsigned long long t0 = wall_clock64();
                bool timeout = false;
                for (unsigned spin = 0;; ++spin) {
                    const bool ok = load_rec(p, tag, nq, val) || !mine;
                    if (__all(ok)) break;
                    if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) {
                        timeout = true;
                        break;
                    }
                }
                if (timeout) {
                    if (lane == 0) atomicOr(&hdr->status, RLVI_ST_TIMEOUT);
                    dead = true;
                }
            }
            TB_STAMP();   // totals in
            const float gm = (lane < Ke && nq > 4) ? val[4] : __builtin_inff();
            if ((RLVI_STAMPS && dbg != nullptr) && b == 0 && round == 0) dbg[900 + lane] = ((unsigned long long)nq << 32) | __float_as_uint(val[4]);
            if (HI_OK && hi_round)
                tj_chain<true, true, HI_OK>(sh.out, Ke, K, val[0], val[1], val[2], val[3], gm, dead, rn_l, shift,
                                            invN, tol, trace, true, xstep, dbg, val[5], val[6]);
            else if (round == 0)
                tj_chain<true>(sh.out, Ke, K, val[0], val[1], val[2], val[3],
                               gm, dead,
                               rn_l, shift, invN, tol, trace, true, xstep, dbg, 0.0f, 0.0f, !wm.warm);
            else
                tj_chain<false>(sh.out, Ke, K, val[0], val[1], val[2], val[3],
                               gm, dead,
                                rn_l, shift, invN, tol, trace, true, xstep, dbg);
        }
        __syncthreads();
        ++tag; ++xstep; ++ptag;
        TB_STAMP();   // recurrence done
        dead = sh.out.dead != 0;
        it = sh.out.res_it;
        r_fin = sh.out.res_rfin;
        const float delta = sh.out.res_delta;
        const bool found = sh.out.res_found != 0;
        if ((RLVI_STAMPS && dbg != nullptr) && b == 0 && tid == 0 && round < 24)
            dbg[64 + round] = ((unsigned long long)((Ke << 8) | it) << 32) | __float_as_uint(delta);
        rn_l = lane < K ? sh.out.nodes[lane] : 1.0f;
        if (lane == 0) rn_l = (float)(0.95 / (1.0 - 0.95));
        Ke = __builtin_amdgcn_readfirstlane(found ? round8(it + 2) : K);      // no stop index among the evaluated nodes: all of them
        if (round == 0) {
            // the true minimum is known now: residuals.sub_(min) (:27, stored by the caller),
            // e = exp(-residuals) (:28)
            gmin = sh.out.res_min;
#pragma unroll
            for (int j = 0; j < E; ++j) ev[j] = expf(-(l[j] - gmin));      // pads stay 0
            if (delta > TJ_ACCEPT && !dead) stage_slice(false);             // (another round will read them)
        }
        if (delta <= TJ_ACCEPT) { accepted = true; break; }
        if (dead) break;
    }
    // every round makes at least one more node exact; anything else is a bug or a non-finite
    // input: report it instead of returning silently wrong posteriors
    if (!accepted && !dead && tid == 0) atomicOr(&hdr->status, RLVI_ST_NOCONV);

    TB_STAMP();   // solved
    if ((RLVI_STAMPS && dbg != nullptr) && b == 0 && tid == 0) dbg[63] = (unsigned long long)dbgi;
    if (b == 0 && tid < WAVE) {
        if (tid == 0) {
            if (out_iters != nullptr) *out_iters = it;
            state->n = (long long)N;
            state->k = K;
            state->shift = gmin;
            state->it = it;
            if (pt != nullptr) pt->dtag = ptag - 1u;
            __hip_atomic_store((gu32 *)&hdr->epoch_base, tag + 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < K) state->nodes[tid] = rn_l;
    }
    TbSolved r;
    r.r_fin = r_fin; r.gmin = gmin; r.it = it; r.accepted = accepted; r.dead = dead;
    r.tag_free = tag;    // rounds used tags below this one; the next launch starts at tag + 2
    return r;
}

}  // namespace rlvi
