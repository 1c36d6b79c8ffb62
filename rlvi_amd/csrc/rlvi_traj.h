// The trajectory E-step (estep_trajb.hip): the warm-start state kept in the workspace and the
// scalar recurrence one wave runs on the gathered per-node totals.
#pragma once
#include "rlvi_coop.h"

namespace rlvi {

constexpr int TJ_MAXK = 64;
constexpr float TJ_ACCEPT = 1e-6f;
#ifndef RLVI_TJ_TRUST
#define RLVI_TJ_TRUST 0.25f
#endif

struct TrajState {
    long long n;
    int k;
    float shift;                 // min residual of the last call (guess of this call's shift)
    float nodes[TJ_MAXK];
    int it;                      // iterations of the last call (how many nodes are worth evaluating)
};
static_assert(sizeof(TrajState) <= WS_TRAJ_BYTES, "warm-start state must fit its workspace region");

// what the recurrence leaves in LDS for the whole workgroup
struct TjOut {
    float nodes[TJ_MAXK];        // corrected trajectory of the latest round
    int dead;
    int res_it;
    int res_found;               // a stop index was found among the evaluated nodes (or Ke == Ka)
    float res_delta, res_rfin, res_min;
    // per-node coefficients of the serial chain, written by the node's lane and read back by ALL lanes one
    // node at a time (an LDS broadcast read: two 16-byte reads per step, asked for one step ahead, instead
    // of six v_readlane whose SGPR results each cost the wave wait states before a vector instruction may
    // use them): {r', a0, b, c, R3, R4, -, -}
    alignas(16) float coef[TJ_MAXK + 2][8];
};

// Neighbour lanes and an affine scan over the 64 lanes without the LDS crossbar (a __shfl is a
// ds_bpermute, ~100 cycles of latency each, and the recurrence is one long dependent chain).
__device__ __forceinline__ float lane_up1(float v) {      // lane i <- lane i-1 (lane 0 <- 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));   // wave_shr:1
}
__device__ __forceinline__ float lane_down1(float v) {    // lane i <- lane i+1 (lane 63 <- 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));   // wave_shl:1
}
// Inclusive scan of the maps x -> sc x + of along the lanes (lane k ends with the composition of the
// maps of lanes 0..k, applied in that order): row_shr steps inside the 16-lane rows, then the row ends
// through v_readlane.
__device__ __forceinline__ void affine_scan(float &sc, float &of) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int li = lane & 15;
#define RLVI_AFF_STEP(CTRL, SH)                                                                  \
    {                                                                                            \
        const float psc = dpp_x<CTRL>(sc), pof = dpp_x<CTRL>(of);                                \
        if (li >= (SH)) { of = fmaf(sc, pof, of); sc *= psc; }                                   \
    }
    RLVI_AFF_STEP(0x111, 1)
    RLVI_AFF_STEP(0x112, 2)
    RLVI_AFF_STEP(0x114, 4)
    RLVI_AFF_STEP(0x118, 8)
#undef RLVI_AFF_STEP
    auto rl = [](float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
    const float s1 = rl(sc, 15), o1 = rl(of, 15), s2 = rl(sc, 31), o2 = rl(of, 31), s3 = rl(sc, 47), o3 = rl(of, 47);
    // prefixes of the rows: P1 = e1, P2 = e2 o e1, P3 = e3 o P2
    const float p2s = s2 * s1, p2o = fmaf(s2, o1, o2);
    const float p3s = s3 * p2s, p3o = fmaf(s3, p2o, o3);
    const int row = lane >> 4;
    const float ps = row == 1 ? s1 : row == 2 ? p2s : p3s;
    const float po = row == 1 ? o1 : row == 2 ? p2o : p3o;
    if (row > 0) { of = fmaf(sc, po, of); sc *= ps; }
}

// Ascending bitonic sort of one key per lane over the 64 lanes of a wave; `src` (initialised to the
// lane id by the caller) ends up as the lane the key of each sorted position came from.  Ties are
// ordered by `src`, so the compare-exchange is a strict total order and both lanes of a pair agree.
__device__ __forceinline__ void wave_sort_keys(float &key, int &src) {
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int k = 2; k <= WAVE; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float ok = __shfl_xor(key, j, WAVE);
            const int os = __shfl_xor(src, j, WAVE);
            const bool want_min = ((lane & j) == 0) == ((lane & k) == 0);
            const bool other_less = ok < key || (ok == key && os < src);
            if (want_min == other_less) { key = ok; src = os; }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The recurrence of train_rlvi.py:30-37 on one wave.  Lane j holds node j: its guess rn_l and the
// totals tS = S(rn), tP = dS/dr(rn), tQ = -1/2 d2S/dr2(rn) (0: first-order correction only),
// tD = D(rn_j, rn_{j-1}) (D_0 against the caller's pi) over ALL
// samples, for j < Ke (the evaluated nodes); Ka >= Ke nodes are kept in the state.  FIRST: the
// sums were taken with e' = exp(-(l - shift)) and gmin (any lane's value is min-reduced here) is
// the true minimum.  Results go to `out` (lane 0 / lanes < Ka); every lane must call.
// HASQ: tQ is meaningful (second-order local model, quintic global model); otherwise first-order /
// cubic.
// ---------------------------------------------------------------------------------------
// HI (first round of estep_trajb.hip): tR3 = sum e^3/(1+re)^4 and tR4 = sum e^4/(1+re)^5 (the third- and
// fourth-order terms of S around the node) are given too; the chain then runs on the fourth-order
// model and may ACCEPT THE ROUND WITHOUT A VERIFICATION ROUND (see below).
template <bool FIRST, bool HASQ = true, bool HI = false>
__device__ __forceinline__ void tj_chain(TjOut &out, int Ke, int Ka, float tS, float tP, float tQ,
                                         float tD,
                                         float gmin, bool dead, float rn_l, float shift, float invN,
                                         float tol, float *trace, bool want_nodes, int xstep,
                                         unsigned long long *dbg, float tR3 = 0.0f, float tR4 = 0.0f,
                                         bool cold = false) {
    const int lane = threadIdx.x & (WAVE - 1);
    const bool has = lane < Ke;
    // RLVI_TJ_DEBUG: where the recurrence wave's time goes (first round of workgroup 0)
#define TJ_STAMP(k) do { if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && xstep == 0 && lane == 0) dbg[990 + (k)] = wall_clock64(); } while (0)
    TJ_STAMP(0);
    float scale = 1.0f;
    if (FIRST) {
        gmin = group_allreduce<WAVE>(gmin, FMin());
        // the sums were taken with e' = exp(-(l - shift)) = e * exp(shift - min): in r-space the
        // evaluated nodes are rn * exp(shift - min); not trustworthy if that factor is extreme
        // or a sum overflowed
        scale = expf(shift - gmin);
    }
    const float rn = rn_l * scale;
    // (a node far below the trajectory may legitimately have S = 0 after fp32 underflow)
    const float finf = __builtin_inff();          // (x < inf is false for +inf and for NaN)
    bool finite = has ? (tS < finf && tP < finf && tD < finf && tQ < finf) : true;
    if (HI && has) finite = finite && tR3 < finf && tR4 < finf;
    const bool round_ok = __all(finite) && scale > 1e-6f && scale < 1e6f && !dead;
    if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && xstep == 0) {
        if (lane < 8) { dbg[104 + 3 * lane] = (unsigned long long)__double_as_longlong((double)tS); dbg[105 + 3 * lane] = (unsigned long long)__double_as_longlong((double)tP); dbg[106 + 3 * lane] = (unsigned long long)__double_as_longlong((double)tQ); }
        if (lane == 0) { dbg[102] = __ballot(finite); dbg[103] = __float_as_uint(scale); }
    }

    // lane-parallel: a0 = mean(pi) at the node, b = d mean / dr, err = ||new - old||_2
    const float a0_l = has ? tS * invN : 0.0f;                                // (:35)
    // (FIRST: tP, tQ are derivatives with respect to r' = r / scale; b and c only steer the
    //  correction, fp32 is plenty)
    const float iscale = FIRST ? __builtin_amdgcn_rcpf(scale) : 1.0f;
    const float b_l = has ? tP * invN * iscale : 0.0f;
    const float c_l = has ? tQ * invN * iscale * iscale : 0.0f;
    // (mean-pi units, like b and c; P2 stays a plain sum: D = h^2 * P2)
    const float r3_l = (HI && has) ? tR3 * invN * iscale * iscale * iscale : 0.0f;
    const float r4_l = (HI && has) ? tR4 * invN * iscale * iscale * iscale * iscale : 0.0f;
    const float err_l = has ? sqrtf(tD) : __builtin_inff();                   // (:33)
    const unsigned long long stopmask = __ballot(has && err_l < tol);                // (:36)
    const int it_now = stopmask ? (int)__builtin_ctzll(stopmask) + 1 : Ke;
    const bool found = stopmask != 0ull || Ke >= Ka;
    const int steps = it_now + 2 < Ke ? it_now + 2 : Ke;      // a little lookahead
    float r = (float)(0.95 / (1.0 - 0.95));
    float rnew_l = rn;
    float avg_l = 0.0f;
    // Verification rounds first, without the serial loop: when the nodes already (almost) satisfy
    // the recurrence, the relative deviation eps_k = (r_k - r'_k)/r'_k of the true chain obeys the
    // LINEAR recurrence  eps_{k+1} = rho_k + s_k eps_k,  rho_k = (g(a0_k) - r'_{k+1})/r'_{k+1},
    // s_k = b_k r'_k / ((1 - a0_k)^2 r'_{k+1})  (second-order terms ~20 eps^2), which is an affine
    // scan over the lanes: six shuffle steps instead of it+2 dependent iterations.
    bool scanned = false;
    if (!FIRST) {
        const float om = 1.0f - a0_l;
        const float iom = __builtin_amdgcn_rcpf(om);
        const float rn_next = __shfl_down(rn, 1, WAVE);
        const float irn = __builtin_amdgcn_rcpf(rn_next);
        const bool act = has && lane + 1 < steps;
        float sc = act ? b_l * rn * iom * iom * irn : 0.0f;            // s_k
        float of = act ? (a0_l * iom - rn_next) * irn : 0.0f;          // rho_k
        if (lane == 0) of = fmaf(sc, (r - rn) * __builtin_amdgcn_rcpf(rn), of);   // eps_0 (r_0 is exact)
#pragma unroll
        for (int sh = 1; sh < WAVE; sh <<= 1) {                        // inclusive scan of x -> sc x + of
            const float psc = __shfl_up(sc, sh, WAVE);
            const float pof = __shfl_up(of, sh, WAVE);
            if (lane >= sh) { of = fmaf(sc, pof, of); sc *= psc; }
        }
        float eps = __shfl_up(of, 1, WAVE);                            // lane k: eps_k
        if (lane == 0) eps = (r - rn) * __builtin_amdgcn_rcpf(rn);
        const float emax = group_allreduce<WAVE>((has && lane < steps) ? fabsf(eps) : 0.0f, FMax());
        if (emax <= 3e-5f) {                                           // (NaN compares false)
            scanned = true;
            rnew_l = fmaf(rn, eps, rn);
            avg_l = fmaf(b_l * rn, eps, a0_l);
        }
    }
    TJ_STAMP(1);   // lane-parallel preparation done
    // serial chain; `step` is wave-uniform, so the per-node values come through v_readlane
    // (SGPR lane select, no LDS):  avg = a0 + b d - c d^2, d = r - r',  r <- avg / (1 - avg).
    // Fast form first (five dependent fp32 operations per step); its steps are then checked
    // lane-parallel against the trust region |r - r'| <= r'/2, 0 < avg < 1, and only a chain
    // that left it (cold or poor guesses) is redone on the global model below.
    // (Measured: 62 ns per step = ~19 clocks per level of the seven-level dependent chain dr -> dr^2 ->
    //  Estrin pair -> avg -> 1 - avg -> rcp -> next dr; the six v_readlane of a step already issue in its
    //  stalls -- writing four steps per loop iteration side by side changed nothing.)
    // (cold: the nodes are the geometric default, not a guess of this trajectory -- the local chain would leave
    //  its trust region at once; its 1.2 us are saved and the global model below takes the round)
    if (!scanned && !cold) {
        if (HI) {
            // Fourth-order chain.  S(r' + d) = S + S' d - Q d^2 + R3 d^3 - R4 d^4 + ...  (alternating for
            // d > 0, terms falling by a factor <= |d|/r').  The arithmetic of a step is what it always was
            // (Estrin: three dependent levels behind dr); its six per-node operands now come through LDS.
            {
                float *cf = out.coef[lane];
                *reinterpret_cast<float4 *>(cf) = make_float4(rn, a0_l, b_l, c_l);
                *reinterpret_cast<float2 *>(cf + 4) = make_float2(r3_l, r4_l);
                if (lane == 0) *reinterpret_cast<float4 *>(out.coef[TJ_MAXK]) = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
            // two register sets, two steps per trip: the operands of step k + 1 are asked for BEFORE the
            // arithmetic of step k, so the LDS latency (~100 clocks) runs beside a step instead of in front of
            // every one (written out by hand: with one set and a copy the compiler rotates the loop and the
            // read lands at the top of the step that needs it).  A lone wave issues one vector instruction per
            // ~6 clocks whatever it depends on (tools/lab/valu_rate.hip), so a step costs its instruction
            // count (52 ns): it carries avg and 1 / (1 - avg) instead of r = avg / (1 - avg), takes
            // dr = avg * inv - r' as ONE fused multiply-add and leaves {avg_k, inv_k} in the node's slot of the
            // operand table -- one LDS store, no compare / selects; the lanes form their own
            // r_k = avg_{k-1} * inv_{k-1} after the loop
            float ap = r, ip = 1.0f;                                   // r_0 = ap * ip exactly
            auto chain_step = [&](int step, const float4 &ca, const float2 &cb) {
                const float dr = fmaf(ap, ip, -ca.x);                  // r_k - r'_k
                const float dr2 = dr * dr;
                const float lo2 = fmaf(dr, ca.z, ca.y), hi2 = fmaf(dr, cb.x, -ca.w);
                const float avg = fmaf(dr2 * dr2, -cb.y, fmaf(dr2, hi2, lo2));
                const float inv = __builtin_amdgcn_rcpf(1.0f - avg);   // (:31)
                *reinterpret_cast<float2 *>(out.coef[step] + 6) = make_float2(avg, inv);
                ap = avg; ip = inv;
            };
            float4 a0v = *reinterpret_cast<const float4 *>(out.coef[0]);
            float2 b0v = *reinterpret_cast<const float2 *>(out.coef[0] + 4);
            int step = 0;
#pragma unroll 1
            for (; step + 1 < steps; step += 2) {
                const float4 a1v = *reinterpret_cast<const float4 *>(out.coef[step + 1]);
                const float2 b1v = *reinterpret_cast<const float2 *>(out.coef[step + 1] + 4);
                __builtin_amdgcn_sched_barrier(0);             // (the reads are issued HERE, ahead of the step)
                chain_step(step, a0v, b0v);
                a0v = *reinterpret_cast<const float4 *>(out.coef[step + 2]);
                b0v = *reinterpret_cast<const float2 *>(out.coef[step + 2] + 4);
                __builtin_amdgcn_sched_barrier(0);
                chain_step(step + 1, a1v, b1v);
            }
            if (step < steps) chain_step(step, a0v, b0v);
            __builtin_amdgcn_wave_barrier();
            if (lane < steps) {
                avg_l = out.coef[lane][6];
                if (lane > 0) {
                    const float2 pv = *reinterpret_cast<const float2 *>(out.coef[lane - 1] + 6);
                    rnew_l = pv.x * pv.y;                              // r_k = avg_{k-1} / (1 - avg_{k-1})
                } else {
                    rnew_l = r;
                }
            }
            r = ap * ip;                                               // (the r after the last step, as before)
        } else {
#pragma unroll 1
            for (int step = 0; step < steps; ++step) {
                const float rns = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rn), step));
                const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0_l), step));
                const float bb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), step));
                const float cc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_l), step));
                if (lane == step) rnew_l = r;
                const float dr = r - rns;
                const float avg = fmaf(dr, fmaf(-cc, dr, bb), a0);
                if (lane == step) avg_l = avg;
                r = avg * __builtin_amdgcn_rcpf(1.0f - avg);                              // (:31)
            }
        }
    }
    TJ_STAMP(2);   // serial chain done
    // (25 % trust region for the local model: inside it one more round finishes -- the bench's
    //  data-to-data drift of 2-7 % stays on this path; beyond it the local model converges one
    //  node per round at worst)
    const bool inside = !cold && (!(has && lane < steps) ||
                        (fabsf(rnew_l - rn) <= RLVI_TJ_TRUST * rn && avg_l > 0.0f && avg_l < 0.999999f));
    if (!__all(inside)) {
        // Cold or poor guesses: the nodes are far from the trajectory, but together they sample
        // S(r) over its whole range.  s(u) = mean(pi) as a function of u = log r is a sum of
        // logistic sigmoids -- smooth and monotone -- so a two-point Hermite interpolant between
        // the evaluated nodes that bracket the current r (values, slopes and, with HASQ, second
        // derivatives: quintic, error ~h^6/46080 |s^(6)| at spacing h = ln 2) gives every step of
        // the recurrence to ~1e-5 whatever the guess was; the next round's local model finishes.
        // Outside the sampled range: the damped local model of the nearest node (+-50 %).
        float key = (has && rn > 0.0f && a0_l == a0_l) ? __logf(rn) : __builtin_inff();
        int src = lane;
        // The nodes of a trajectory fall with the lane (a cold start's geometric guesses always do): the ascending
        // order is then the reversal of the evaluated lanes -- one permute instead of the 21 compare-exchange
        // stages of the bitonic sort (42 ds_bpermute round trips, ~2 us on a lone wave).  Same order, same bits.
        {
            const float kprev = lane_up1(key);
            const bool falls = !has || (key < __builtin_inff() && (lane == 0 || key < kprev));
            if (__all(falls)) {
                src = lane < Ke ? Ke - 1 - lane : lane;
                key = __shfl(key, src, WAVE);
            } else {
                wave_sort_keys(key, src);
            }
        }
        const float rn_s = __shfl(rn, src, WAVE);
        const float s0_s = __shfl(a0_l, src, WAVE);
        const float b_s = __shfl(b_l, src, WAVE);
        const float c_s = __shfl(c_l, src, WAVE);
        const float s1_s = rn_s * b_s;                                   // ds/du
        const float s2_s = fmaf(-2.0f * rn_s * rn_s, c_s, s1_s);         // d2s/du2
        const unsigned long long vmask = __ballot(key < __builtin_inff());
        const int nv = (int)__popcll(vmask);
        auto rl = [](float v, int p) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), p)); };
        // The interpolant of every interval, once and lane-parallel (round 4): sorted position p holds the interval
        // [node p-1, node p] as the six monomial coefficients of its Hermite polynomial in t = (u - u_a) / h, with
        // u_a and 1 / h beside them, in the operand table -- a step of the chain below is then a logarithm, a ballot,
        // two 16-byte LDS reads and a Horner scheme instead of ten v_readlane (each a scalar result the next vector
        // instruction has to wait for) and the five basis polynomials: 0.26 -> 0.1 us per step of a cold start.
        {
            const float ua = lane_up1(key), fa = lane_up1(s0_s), da = lane_up1(s1_s), ea = lane_up1(s2_s);
            const float h = key - ua;
            const bool okiv = lane >= 1 && lane < nv && h > 1e-5f;
            const float df = s0_s - fa, g0 = h * da, g1 = h * s1_s;
            float c2, c3, c4 = 0.0f, c5 = 0.0f;
            if (HASQ) {
                const float q0 = h * h * ea, q1 = h * h * s2_s;
                c2 = 0.5f * q0;
                c3 = fmaf(10.0f, df, fmaf(-6.0f, g0, fmaf(-4.0f, g1, fmaf(-1.5f, q0, 0.5f * q1))));
                c4 = fmaf(-15.0f, df, fmaf(8.0f, g0, fmaf(7.0f, g1, fmaf(1.5f, q0, -q1))));
                c5 = fmaf(6.0f, df, fmaf(-3.0f, g0, fmaf(-3.0f, g1, fmaf(-0.5f, q0, 0.5f * q1))));
            } else {
                c2 = fmaf(3.0f, df, fmaf(-2.0f, g0, -g1));
                c3 = fmaf(-2.0f, df, g0 + g1);
            }
            float *cf = out.coef[lane];
            *reinterpret_cast<float4 *>(cf) = make_float4(fa, g0, c2, c3);
            *reinterpret_cast<float4 *>(cf + 4) = make_float4(c4, c5, ua, okiv ? __builtin_amdgcn_rcpf(h) : -1.0f);
        }
        __builtin_amdgcn_wave_barrier();
        r = (float)(0.95 / (1.0 - 0.95));
#pragma unroll 1
        for (int step = 0; step < steps; ++step) {
            if (lane == step) rnew_l = r;
            const float u = __logf(r);
            const unsigned long long ge = __ballot(key >= u) & vmask;
            const int p_hi = ge ? (int)__builtin_ctzll(ge) : nv;          // first node at or above r
            float avg;
            bool done = false;
            if (p_hi > 0 && p_hi < nv) {
                const float4 ca = *reinterpret_cast<const float4 *>(out.coef[p_hi]);
                const float4 cb = *reinterpret_cast<const float4 *>(out.coef[p_hi] + 4);
                if (cb.w > 0.0f) {
                    const float t = (u - cb.z) * cb.w;
                    avg = fmaf(t, fmaf(t, fmaf(t, fmaf(t, fmaf(t, cb.y, cb.x), ca.w), ca.z), ca.y), ca.x);
                    done = true;
                }
            }
            if (!done) {
                // no bracket (or two coinciding nodes): damped local model of the nearest node
                const int p = p_hi < nv ? p_hi : nv - 1;
                const float rns = rl(rn_s, p), a0 = rl(s0_s, p), bb = rl(b_s, p), cc = rl(c_s, p);
                const float hh = 0.5f * rns;
                const float d = fmaxf(fminf(r - rns, hh), -hh);
                avg = fmaf(d, fmaf(-cc, d, bb), a0);
            }
            avg = fminf(fmaxf(avg, 0.0f), 0.999999f);
            if (lane == step) avg_l = avg;
            r = fmaxf(avg * __builtin_amdgcn_rcpf(1.0f - avg), 1e-30f);              // (:31)
        }
    }
    // nodes beyond the lookahead: a fresh geometric tail from the last corrected node
    {
        const float last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rnew_l), steps - 1));
        if (lane < Ka && lane >= steps) rnew_l = last * exp2f(-(float)(lane - steps + 1));
    }
    float d_l = (has && lane < it_now) ? fabsf(rnew_l - rn) * __builtin_amdgcn_rcpf(rn) : 0.0f;
    float delta_w = group_allreduce<WAVE>(d_l, FMax());
    // ---- Accept WITHOUT a verification round (HI, the local model held everywhere).
    // What a verification round would deliver is the step errors err_k = ||pi(r_k) - pi(r_{k-1})||
    // at the true nodes, i.e. the stop index.  This round measured D'_k at the GUESSED pair
    // (r'_k, r'_{k-1}).  Write D = h^2 * P(rbar), h the step and rbar the midpoint of a pair: P is a
    // smooth function of the pair's position (it is sum e^2/(1+re)^4 for small h/r and carries the
    // curvature of pi otherwise), so
    //     err_k = err'_k * |h_k / h'_k| * (rbar_k / rbar'_k)^(sigma/2),   sigma = dlnP/dlnrbar
    // (a secant through the measured pairs) is good to second order in how far the pair moved.
    // What it needs is h_k, i.e. the corrected nodes, whose relative error E_k is BOUNDED here: the
    // model's remainder at node k is at most 1.34 R4 d^4 |d|/r' (alternating series with falling
    // terms for d > 0, a geometric tail of ratio <= 1/4 for d < 0) plus the fp32 floor of the sums,
    // it enters r_{k+1} through 1/(1-a)^2 and propagates with the chain's own factor s_k -- an
    // affine scan over the lanes.  Every stop test up to the stop index has to clear tol by a band of
    // 3 x sqrt2 E r/|h| + 2 x {2 mv^2 + |sigma dln rbar|/4 + 0.5 %}; anything inside a band: no
    // accept, the verification round runs.
    TJ_STAMP(3);   // trust check / tail done
    int it_acc = 0;
    bool accept_now = false;
    if (HI && FIRST && __all(inside) && !scanned && round_ok && trace == nullptr && steps >= 2) {
        const bool live = has && lane < steps;
        const float dk = rnew_l - rn;                                        // d at this node
        const float om = 1.0f - avg_l;
        const float iom2 = __builtin_amdgcn_rcpf(om * om);
        const float rnext = lane_down1(rnew_l);
        const float irn = __builtin_amdgcn_rcpf(fmaxf(rnext, 1e-30f));
        const float adk = fabsf(dk);
        // remainder of mean(pi) at this node (+ the fp32 floor of the sums), as relative error of r_{k+1}
        const float rem = 1.34f * r4_l * dk * dk * dk * dk * adk * __builtin_amdgcn_rcpf(rn) + 1e-7f * a0_l;
        const bool act = live && lane + 1 < steps;
        float of = act ? rem * iom2 * irn : 0.0f;                            // rho_k
        float sc = act ? fminf(fabsf(b_l) * rnew_l * iom2 * irn, 1.0f) : 0.0f;   // s_k
        if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && xstep == 0)
            dbg[828 + lane] = ((unsigned long long)__float_as_uint(sc) << 32) | __float_as_uint(of);
        affine_scan(sc, of);                                                 // x -> sc x + of, inclusive
        float Ek = lane_up1(of);                                             // bound on |r_k - true| / r_k
        if (lane == 0) Ek = 0.0f;                                            // r_0 is exact
        Ek += 1.2e-7f;                                                       // the nodes are fp32 numbers
        const float Ekm = lane_up1(Ek);
        // the measured pairs: step, midpoint, P = D'/h'^2
        const float rp_new = lane_up1(rnew_l);
        const float rp_old = lane_up1(rn);
        const float h = rnew_l - rp_new, hq = rn - rp_old;                   // corrected / guessed step
        const float rbar = 0.5f * (rnew_l + rp_new), rbarq = 0.5f * (rn + rp_old);
        const float lrq = __logf(fmaxf(rbarq, 1e-30f));
        const float lpq = 2.0f * (__logf(fmaxf(err_l, 1e-30f)) - __logf(fmaxf(fabsf(hq), 1e-30f)));   // ln P'
        const bool pair = live && lane >= 1;
        // slope of ln P against ln rbar: secant to the previous pair where that is at least 2 % away,
        // else (pairs crowding at a fixed point) the secant from the last pair to the nearest one that is
        const float lrq_p = lane_up1(lrq), lpq_p = lane_up1(lpq);
        const float lr_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lrq), steps - 1));
        const float lp_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lpq), steps - 1));
        const unsigned long long far = __ballot(pair && fabsf(lrq - lr_last) >= 0.02f);
        const int jstar = far ? 63 - (int)__builtin_clzll(far) : 1;
        const float lr_j = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lrq), jstar));
        const float lp_j = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lpq), jstar));
        const float slope_c = far ? (lp_j - lp_last) * __builtin_amdgcn_rcpf(lr_j - lr_last) : 0.0f;
        const float slope_own = (lane >= 2 && fabsf(lrq - lrq_p) >= 0.02f) ? (lpq - lpq_p) * __builtin_amdgcn_rcpf(lrq - lrq_p)
                                                                           : slope_c;
        float slope = fminf(fmaxf(slope_own, -6.0f), 2.0f);
        const float dl = __logf(fmaxf(rbar, 1e-30f)) - lrq;                  // how far the pair moved
        const float ihq = __builtin_amdgcn_rcpf(fmaxf(fabsf(hq), 1e-30f));
        const float err_hat = err_l * fabsf(h) * ihq * __expf(0.5f * slope * dl);
        const float mvk = adk * __builtin_amdgcn_rcpf(rn);
        const float mv = fmaxf(mvk, lane_up1(mvk));                          // relative move of the pair's nodes
        const float node_term = 1.4143f * fmaxf(Ek, Ekm) * rbar * __builtin_amdgcn_rcpf(fmaxf(fabsf(h), 1e-30f));
        float band = 3.0f * node_term + 2.0f * (2.0f * mv * mv + 0.25f * fabsf(slope * dl) + 0.005f);
        float err_e = err_hat;
        if (lane == 0) { err_e = err_l; band = 0.0f; }                       // node 0 and the caller's pi are exact
        const bool clear = !live || (band < 0.5f && fabsf(err_e - tol) > band * err_e);
        const unsigned long long stop_e = __ballot(live && err_e < tol);
        const unsigned long long unclear = __ballot(!clear);
        if (stop_e != 0ull) {
            const int ks = (int)__builtin_ctzll(stop_e);                      // estimated stop index
            const unsigned long long upto = ks >= 63 ? ~0ull : ((2ull << ks) - 1ull);
            if ((unclear & upto) == 0ull && ks + 1 <= steps) { accept_now = true; it_acc = ks + 1; }
        }
        if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && xstep == 0) {
            dbg[700 + lane] = ((unsigned long long)__float_as_uint(err_e) << 32) | __float_as_uint(band);
            dbg[764 + lane] = ((unsigned long long)__float_as_uint(Ek) << 32) | __float_as_uint(slope);
            if (lane == 0) dbg[699] = ((unsigned long long)(accept_now ? 1 : 0) << 32) | (unsigned)it_acc;
        }
    }
    TJ_STAMP(4);   // acceptance test done
    // Early accept: with nodes off by delta the corrected r are good to 0.25 delta^2, and the
    // errors (evaluated AT the nodes) to about delta*(r_k + r_{k-1})/|r_k - r_{k-1}| relative.
    // If every stop test up to `it` clears tol by 8x that margin, the stop index cannot change
    // any more, and neither can pi: no verification round needed.  (Not when the caller asked
    // for the error trace: that wants the errors themselves.)
    const float rp_l = lane_up1(rn);
    if (delta_w > TJ_ACCEPT && delta_w <= 1e-3f && trace == nullptr) {
        float u = 0.0f;
        if (has && lane < it_now && lane > 0) {
            const float gap = fabsf(rn - rp_l);
            u = 8.0f * delta_w * (rn + rp_l) * __builtin_amdgcn_rcpf(fmaxf(gap, 1e-30f));
        }
        if (lane == 0) u = 128.0f * delta_w;            // D_0 is taken against the caller's pi
        const bool unsafe = has && lane < it_now && fabsf(err_l - tol) <= u * err_l;
        if (__ballot(unsafe) == 0ull) delta_w = 0.0f;
    }
    if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && xstep < 4)
        dbg[400 + xstep * 64 + lane] = ((unsigned long long)__float_as_uint(rn) << 32) | __float_as_uint(rnew_l);
    if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && xstep == 0) {
        dbg[130 + lane] = ((unsigned long long)__float_as_uint(rn) << 32) | __float_as_uint(rnew_l);
        if (lane == 0) { dbg[128] = round_ok; dbg[129] = ((unsigned long long)it_now << 32) | __float_as_uint(delta_w); }
    }
    // guesses stay where sums in fp32 cannot underflow; an invalid round restarts from the
    // geometric cold guess instead of re-evaluating the nodes that broke it; a round that did not
    // reach the stop index (too few nodes evaluated) is never accepted
    rnew_l = fminf(fmaxf(rnew_l, 1e-30f), 1e30f);
    if (!found) delta_w = __builtin_inff();
    if (!round_ok) { delta_w = __builtin_inff(); rnew_l = 19.0f * exp2f(-(float)lane); }
    if (lane < Ka && want_nodes) out.nodes[lane] = rnew_l;
    if (trace != nullptr && blockIdx.x == 0 && lane < it_now && round_ok && found) {
        trace[2 * lane] = err_l;
        trace[2 * lane + 1] = avg_l;
    }
    const int it_out = accept_now ? it_acc : it_now;
    if (accept_now) delta_w = 0.0f;
    const float rfin_w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rnew_l), it_out - 1));
    if (lane == 0) {
        out.res_it = it_out;
        out.res_found = ((found || accept_now) && round_ok) ? 1 : 0;
        out.res_delta = delta_w;
        out.res_rfin = rfin_w;
        if (FIRST) out.res_min = gmin;
        out.dead = dead ? 1 : 0;
    }
    TJ_STAMP(5);
#undef TJ_STAMP
}

}  // namespace rlvi
