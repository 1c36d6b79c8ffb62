// Type-II-error threshold + truncation + selection mask, without sorting.
//
// Replaces false_negative_criterion (deep-learning/methods/train_rlvi.py:41-49: sum, sort
// descending, cumsum, count, gather), `threshold = max(threshold, ...)` (:102), the truncation
// `weights[weights < threshold] = 0` (:103) and the keep mask `weights > threshold` (main.py:343).
//
// The reference needs s[count-1] where s = sort_desc(pi), F_k = fl32(sum_{j<=k} (1-s_j)) (torch
// CPU cumsum: fp64 running sum, every prefix rounded to fp32) and count = #{k : F_k <= beta}.
// F is monotone, so count is a position on the sorted order and can be found by BISECTION ON THE
// ORDER-PRESERVING KEY of pi with the predicate fl32(S(c)) <= beta, S(c) = sum over {pi_i with
// key >= c} of (1-pi_i).  For pi in [0,1] every (1-pi_i) is a multiple of 2^-24, so fp64 sums of
// up to 2^29 of them are EXACT in any order: S(c) equals the reference's sequential fp64 prefix
// bit for bit, the predicate is evaluated on identical numbers, and the selected position --
// hence the boolean mask -- is bit-exact.  Ties are resolved arithmetically (multiplicity of the
// crossing key).  ~32 block-wide reductions by one 1024-thread workgroup that holds the keys in
// registers; latency-bound (N*4 B <= a few hundred KB), run once per epoch.
#include <stdlib.h>

#include "rlvi_coop.h"

namespace rlvi {

constexpr int THR_BLOCK = 1024;
constexpr int THR_NW = THR_BLOCK / WAVE;

struct Red3 { double s; unsigned long long a, b; };

// Block-wide reduce of {sum (fp64), min (u64), max (u64)}; every thread gets the result.
__device__ __forceinline__ Red3 block_reduce3(double s, unsigned long long mn,
                                              unsigned long long mx) {
    __shared__ double sh_s[THR_NW];
    __shared__ unsigned long long sh_a[THR_NW], sh_b[THR_NW];
    __shared__ Red3 sh_out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s = wave_sum(s);
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) { sh_s[wave] = s; sh_a[wave] = mn; sh_b[wave] = mx; }
    __syncthreads();
    if (wave == 0) {
        double ts = lane < THR_NW ? sh_s[lane] : 0.0;
        unsigned long long ta = lane < THR_NW ? sh_a[lane] : ~0ull;
        unsigned long long tb = lane < THR_NW ? sh_b[lane] : 0ull;
        ts = wave_sum(ts);
        ta = wave_min(ta);
        tb = wave_max(tb);
        if (lane == 0) { sh_out.s = ts; sh_out.a = ta; sh_out.b = tb; }
    }
    __syncthreads();
    const Red3 r = sh_out;
    __syncthreads();   // sh_out is rewritten by the next call
    return r;
}

// E > 0: keys live in registers (N <= 1024*E).  E == 0: keys are re-read from memory per pass.
template <int E>
struct Keys {
    uint32_t k[E > 0 ? E : 1];
    const float *w;
    int64_t N;
    int cnt;

    __device__ __forceinline__ void load(const float *w_, int64_t N_) {
        w = w_;
        N = N_;
        if (E > 0) {
            cnt = 0;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int64_t i = (int64_t)j * THR_BLOCK + threadIdx.x;
                if (i < N) { k[j] = f32_key(w[i]); cnt = j + 1; } else k[j] = 0;
            }
        }
    }
    template <class Fn>
    __device__ __forceinline__ void for_each(Fn fn) const {
        if (E > 0) {
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (j < cnt) fn(k[j], (int64_t)j * THR_BLOCK + threadIdx.x);
        } else {
            for (int64_t i = threadIdx.x; i < N; i += THR_BLOCK) fn(f32_key(w[i]), i);
        }
    }
};

__device__ __forceinline__ double one_minus(uint32_t key) {
    return (double)(1.0f - key_f32(key));   // fp32 subtraction, as `1 - sorted_weights`
}

template <int E, bool TRUNC>
__global__ __launch_bounds__(THR_BLOCK) void threshold_kernel(float *__restrict__ w, int64_t N,
                                                              float alpha,
                                                              float *__restrict__ thr_io,
                                                              uint8_t *__restrict__ mask,
                                                              int64_t *__restrict__ kept_out,
                                                              const int32_t *__restrict__ only_if) {
    if (only_if != nullptr && *only_if == 0) return;     // the fast form already did the work
    Keys<E> keys;
    keys.load(w, N);

    // beta = alpha * sum(1 - w)   (train_rlvi.py:43-44); global min / max key
    double tot = 0.0;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    keys.for_each([&](uint32_t k, int64_t) {
        tot += one_minus(k);
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    });
    Red3 r = block_reduce3(tot, kmin, kmax);
    const float beta = (float)r.s * alpha;
    kmin = r.a;
    kmax = r.b;

    // smallest c in [kmin, kmax+1] with fl32(S(c)) <= beta
    unsigned long long lo = kmin, hi = kmax + 1ull;
    const bool any_ok = 0.0f <= beta;           // S(kmax+1) = 0
    while (any_ok && lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        double s = 0.0;
        keys.for_each([&](uint32_t k, int64_t) { s += (k >= mid) ? one_minus(k) : 0.0; });
        r = block_reduce3(s, 0ull, 0ull);
        if ((float)r.s <= beta) hi = mid; else lo = mid + 1ull;
    }
    const unsigned long long cstar = lo;

    // elements with key >= c* are all inside the prefix; find the next key below and the
    // smallest key inside
    double s_in = 0.0;
    unsigned long long below = 0ull, above = ~0ull;   // max{k < c*}+1 (0 = none), min{k >= c*}
    double n_in = 0.0;
    keys.for_each([&](uint32_t k, int64_t) {
        if (any_ok && k >= cstar) {
            s_in += one_minus(k);
            n_in += 1.0;
            above = k < above ? k : above;
        } else {
            below = (k + 1ull) > below ? (k + 1ull) : below;
        }
    });
    r = block_reduce3(s_in, above, below);
    const double S = r.s;
    above = r.a;
    below = r.b;
    r = block_reduce3(n_in, 0ull, 0ull);
    const double cnt_in = r.s;

    float thr;
    if (below == 0ull) {
        // every element is inside: count = N, threshold = smallest weight
        thr = key_f32((uint32_t)kmin);
    } else {
        const uint32_t v = (uint32_t)(below - 1ull);
        double mult = 0.0;
        keys.for_each([&](uint32_t k, int64_t) { mult += (k == v) ? 1.0 : 0.0; });
        r = block_reduce3(mult, 0ull, 0ull);
        // j = #{i in 1..mult : fl32(S + i*t) <= beta}; monotone in i -> binary search
        const double t = one_minus(v);
        long long jl = 0, jh = (long long)r.s;   // P(jl) true, P(jh) false (minimality of c*)
        if (!any_ok) jh = 0;
        while (jh - jl > 1) {
            const long long jm = jl + ((jh - jl) >> 1);
            if ((float)(S + (double)jm * t) <= beta) jl = jm; else jh = jm;
        }
        const long long j = any_ok ? jl : 0;
        if (cnt_in + (double)j == 0.0) thr = key_f32((uint32_t)kmin);   // last_index = -1 wraps
        else if (j >= 1) thr = key_f32(v);
        else thr = key_f32((uint32_t)above);
    }

    if (TRUNC) {
        const float prev = *thr_io;
        if (!(thr > prev)) thr = prev;            // threshold = max(threshold, criterion)  (:102)
    }
    __syncthreads();
    if (threadIdx.x == 0) *thr_io = thr;

    if (TRUNC) {
        double kept = 0.0;
        keys.for_each([&](uint32_t k, int64_t i) {
            float x = key_f32(k);
            if (x < thr) { x = 0.0f; w[i] = 0.0f; }          // :103
            const bool m = x > thr;                          // main.py:343
            if (mask != nullptr) mask[i] = m ? 1 : 0;
            kept += m ? 1.0 : 0.0;
        });
        r = block_reduce3(kept, 0ull, 0ull);
        if (threadIdx.x == 0 && kept_out != nullptr) *kept_out = (int64_t)r.s;
    }
}

// ---------------------------------------------------------------------------------------
// Fast form for weights in [0, 1] (what the E-step produces): (1 - pi) * 2^24 is an exact
// integer <= 2^24, so the prefix sums are plain integer sums (u32 per thread, u64 across the
// workgroup) -- six 32-bit VALU operations per key and bisection round instead of fp64 adds.
// The first pass checks the range; outside [0, 1] the generic kernel above is used instead
// (signalled through *fallback).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, int tag) {
    __shared__ unsigned long long sh[2][THR_NW];
    __shared__ unsigned long long out[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) sh[tag & 1][wave] = v;
    __syncthreads();
    if (wave == 0) {
        unsigned long long t = lane < THR_NW ? sh[tag & 1][lane] : 0ull;
        t = wave_sum(t);
        if (lane == 0) out[tag & 1] = t;
    }
    __syncthreads();
    return out[tag & 1];      // double-buffered by `tag`: the next call may start before all read
}

__device__ __forceinline__ uint32_t one_minus_u(float p) {
    return (uint32_t)__float2uint_rn((1.0f - p) * 16777216.0f);
}

template <int E, bool TRUNC>
__global__ __launch_bounds__(THR_BLOCK) void threshold_fast_kernel(float *__restrict__ w, int64_t N,
                                                                   float alpha,
                                                                   float *__restrict__ thr_io,
                                                                   uint8_t *__restrict__ mask,
                                                                   int64_t *__restrict__ kept_out,
                                                                   int32_t *__restrict__ fallback) {
    float p[E];
    int cnt = 0;
    unsigned long long tot = 0ull;
    float pmin = __builtin_inff(), pmax = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = (int64_t)j * THR_BLOCK + threadIdx.x;
        if (i < N) {
            p[j] = w[i];
            cnt = j + 1;
            pmin = fminf(pmin, p[j]);
            pmax = fmaxf(pmax, p[j]);
            tot += one_minus_u(p[j]);
        } else {
            p[j] = -1.0f;          // never >= a candidate in [0, 1]
        }
    }
    Red3 r = block_reduce3((double)0.0, (unsigned long long)f32_key(pmin), (unsigned long long)f32_key(pmax));
    const float gmin = key_f32((uint32_t)r.a), gmax = key_f32((uint32_t)r.b);
    if (!(gmin >= 0.0f && gmax <= 1.0f) || (__float_as_uint(gmin) >> 31)) {   // NaN, -0.0 too
        if (threadIdx.x == 0) *fallback = 1;
        return;
    }
    int tag = 0;
    const unsigned long long total = block_sum_u64(tot, tag++);
    const float beta = (float)((double)total * (1.0 / 16777216.0)) * alpha;     // (:43-44)
    auto pred = [&](unsigned long long s_int) {       // fl32(prefix) <= beta   (:46-47)
        return (float)((double)s_int * (1.0 / 16777216.0)) <= beta;
    };

    // smallest candidate value c (as an fp32 bit pattern; weights >= 0 so bits are ordered) in
    // [gmin, next(gmax)] with  sum_{pi >= c} (1 - pi)  <= beta
    unsigned long long lo = __float_as_uint(gmin), hi = (unsigned long long)__float_as_uint(gmax) + 1ull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        const float c = __uint_as_float((uint32_t)mid);
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) s += (p[j] >= c) ? one_minus_u(p[j]) : 0u;
        if (pred(block_sum_u64((unsigned long long)s, tag++))) hi = mid; else lo = mid + 1ull;
    }
    // c* = lo.  lo == bits(gmax)+1 means even the top group alone does not fit.
    const bool top_empty = lo > (unsigned long long)__float_as_uint(gmax);
    const float cstar = top_empty ? __builtin_inff() : __uint_as_float((uint32_t)lo);
    uint32_t s_in = 0, n_in = 0;
    float below = -1.0f, above = __builtin_inff();       // max{pi < c*}, min{pi >= c*}
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (j < cnt) {
            if (p[j] >= cstar) { s_in += one_minus_u(p[j]); ++n_in; above = fminf(above, p[j]); }
            else below = fmaxf(below, p[j]);
        }
    }
    const unsigned long long S = block_sum_u64((unsigned long long)s_in, tag++);
    const unsigned long long cnt_in = block_sum_u64((unsigned long long)n_in, tag++);
    r = block_reduce3(0.0, (unsigned long long)f32_key(above), (unsigned long long)f32_key(below));
    above = key_f32((uint32_t)r.a);
    below = key_f32((uint32_t)r.b);

    float thr;
    if (below < 0.0f) {
        thr = gmin;                                   // every element inside: count = N
    } else {
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) m += (j < cnt && p[j] == below) ? 1u : 0u;
        const unsigned long long mult = block_sum_u64((unsigned long long)m, tag++);
        const unsigned long long t = one_minus_u(below);
        long long jl = 0, jh = (long long)mult;       // pred(jl) true, pred(jh) false
        while (jh - jl > 1) {
            const long long jm = jl + ((jh - jl) >> 1);
            if (pred(S + (unsigned long long)jm * t)) jl = jm; else jh = jm;
        }
        const long long j = pred(S) ? jl : 0;         // S = 0 (empty top) always fits: beta >= 0
        if (cnt_in + (unsigned long long)j == 0ull) thr = gmin;     // last_index = -1 wraps (:47-48)
        else if (j >= 1) thr = below;
        else thr = above;
    }
    if (TRUNC) {
        const float prev = *thr_io;
        if (!(thr > prev)) thr = prev;                // threshold = max(threshold, criterion) (:102)
    }
    __syncthreads();
    if (threadIdx.x == 0) *thr_io = thr;
    if (TRUNC) {
        uint32_t kept = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t i = (int64_t)j * THR_BLOCK + threadIdx.x;
            if (j < cnt) {
                float x = p[j];
                if (x < thr) { x = 0.0f; w[i] = 0.0f; }      // :103
                const bool mk = x > thr;                     // main.py:343
                if (mask != nullptr) mask[i] = mk ? 1 : 0;
                kept += mk ? 1u : 0u;
            }
        }
        const unsigned long long k = block_sum_u64((unsigned long long)kept, tag++);
        if (threadIdx.x == 0 && kept_out != nullptr) *kept_out = (int64_t)k;
    }
}


// ---------------------------------------------------------------------------------------
// Cooperative form for populations beyond one workgroup's registers (N > 65 536: Food-101's
// 75 750, web-scale label-noise sets of ~1e6): 240 workgroups of 256 threads hold N/240 keys each
// in registers and run the SAME search with every block-wide reduction followed by one record
// exchange (rlvi_coop.h: sums / min / max formed in a fixed order, identical on every workgroup;
// the fp64 sums of multiples of 2^-24 are exact, so the order does not matter anyway).  An
// exchange carries two values, so the search is ternary: two candidate keys per step, 19-20 steps
// for the 2^30 candidate patterns instead of 30.  ~25 exchanges of ~2.6 us whatever N is, against
// 0.39 ms (N = 75 750) ... 4.8 ms (N = 1e6) for one workgroup re-reading the keys from L2.
// ---------------------------------------------------------------------------------------
constexpr int THC_BLOCK = 256;
constexpr int THC_G = 240;

template <int E, bool TRUNC>
__global__ __launch_bounds__(THC_BLOCK) void threshold_coop_kernel(float *__restrict__ w, int64_t N,
                                                                   float alpha,
                                                                   float *__restrict__ thr_io,
                                                                   uint8_t *__restrict__ mask,
                                                                   int64_t *__restrict__ kept_out,
                                                                   void *ws) {
    Coop<THC_BLOCK> coop;
    coop.init(ws, (int)gridDim.x);
    // (read before the first exchange: workgroup 0 overwrites it after the last one)
    const float prev = TRUNC ? *thr_io : 0.0f;
    const int64_t L = (N + gridDim.x - 1) / gridDim.x;
    const int64_t lo_i = (int64_t)blockIdx.x * L;
    const int64_t hi_i = lo_i + L < N ? lo_i + L : N;
    uint32_t k[E];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = lo_i + threadIdx.x + (int64_t)j * THC_BLOCK;
        if (i < hi_i) { k[j] = f32_key(w[i]); cnt = j + 1; } else k[j] = 0;
    }
    auto for_each = [&](auto fn) {
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (j < cnt) fn(k[j], lo_i + threadIdx.x + (int64_t)j * THC_BLOCK);
    };

    // beta = alpha * sum(1 - w)   (train_rlvi.py:43-44); global min / max key
    double tot = 0.0, dmin = __builtin_inf(), dmax = -__builtin_inf(), zero = 0.0;
    for_each([&](uint32_t key, int64_t) {
        tot += one_minus(key);
        dmin = (double)key < dmin ? (double)key : dmin;
        dmax = (double)key > dmax ? (double)key : dmax;
    });
    coop.template allreduce2<OpSum, OpMin>(tot, dmin);
    coop.template allreduce2<OpMax, OpSum>(dmax, zero);
    const float beta = (float)tot * alpha;
    const unsigned long long kmin = (unsigned long long)dmin, kmax = (unsigned long long)dmax;

    // smallest c in [kmin, kmax+1] with fl32(S(c)) <= beta; invariant: the predicate holds at hi
    unsigned long long lo = kmin, hi = kmax + 1ull;
    const bool any_ok = 0.0f <= beta;           // S(kmax+1) = 0
    while (any_ok && lo < hi && !coop.dead) {
        const unsigned long long span = hi - lo;
        const unsigned long long m1 = span >= 3 ? lo + span / 3 : lo + (span >> 1);
        const unsigned long long m2 = span >= 3 ? lo + 2 * (span / 3) + 1 : hi;     // m1 < m2 <= hi
        double s1 = 0.0, s2 = 0.0;
        for_each([&](uint32_t key, int64_t) {
            const double t = one_minus(key);
            s1 += (key >= m1) ? t : 0.0;
            s2 += (key >= m2) ? t : 0.0;
        });
        coop.template allreduce2<OpSum, OpSum>(s1, s2);
        if ((float)s1 <= beta) hi = m1;
        else if (m2 >= hi || (float)s2 <= beta) { lo = m1 + 1ull; hi = m2 < hi ? m2 : hi; }
        else lo = m2 + 1ull;
    }
    const unsigned long long cstar = lo;

    double s_in = 0.0, n_in = 0.0, above = __builtin_inf(), below = -1.0;   // min{k >= c*}, max{k < c*}
    for_each([&](uint32_t key, int64_t) {
        if (any_ok && key >= cstar) {
            s_in += one_minus(key);
            n_in += 1.0;
            above = (double)key < above ? (double)key : above;
        } else {
            below = (double)key > below ? (double)key : below;
        }
    });
    coop.template allreduce2<OpSum, OpSum>(s_in, n_in);
    coop.template allreduce2<OpMin, OpMax>(above, below);
    const double S = s_in, cnt_in = n_in;

    float thr;
    if (below < 0.0) {
        thr = key_f32((uint32_t)kmin);           // every element is inside: count = N
    } else {
        const uint32_t v = (uint32_t)below;
        double mult = 0.0;
        zero = 0.0;
        for_each([&](uint32_t key, int64_t) { mult += (key == v) ? 1.0 : 0.0; });
        coop.template allreduce2<OpSum, OpSum>(mult, zero);
        // j = #{i in 1..mult : fl32(S + i*t) <= beta}; monotone in i -> binary search
        const double t = one_minus(v);
        long long jl = 0, jh = (long long)mult;   // P(jl) true, P(jh) false (minimality of c*)
        if (!any_ok) jh = 0;
        while (jh - jl > 1) {
            const long long jm = jl + ((jh - jl) >> 1);
            if ((float)(S + (double)jm * t) <= beta) jl = jm; else jh = jm;
        }
        const long long j = any_ok ? jl : 0;
        if (cnt_in + (double)j == 0.0) thr = key_f32((uint32_t)kmin);   // last_index = -1 wraps
        else if (j >= 1) thr = key_f32(v);
        else thr = key_f32((uint32_t)above);
    }
    if (TRUNC && !(thr > prev)) thr = prev;       // threshold = max(threshold, criterion)  (:102)

    double kept = 0.0;
    if (TRUNC) {
        for_each([&](uint32_t key, int64_t i) {
            float x = key_f32(key);
            if (x < thr) { x = 0.0f; w[i] = 0.0f; }          // :103
            const bool m = x > thr;                          // main.py:343
            if (mask != nullptr) mask[i] = m ? 1 : 0;
            kept += m ? 1.0 : 0.0;
        });
        zero = 0.0;
        coop.template allreduce2<OpSum, OpSum>(kept, zero);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *thr_io = thr;
        if (TRUNC && kept_out != nullptr) *kept_out = (int64_t)kept;
        coop.finish(ws);
    }
}

__global__ __launch_bounds__(256) void truncate_kernel(float *__restrict__ w, int64_t N,
                                                       const float *__restrict__ thr_p,
                                                       uint8_t *__restrict__ mask) {
    const float thr = *thr_p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        float x = w[i];
        if (x < thr) { x = 0.0f; w[i] = 0.0f; }
        if (mask != nullptr) mask[i] = x > thr ? 1 : 0;
    }
}

// N <= 16 384: one workgroup, the integer fast form with the weights in registers (16 per thread,
// 40 us); it hands over to the generic fp64 form (any range) by setting a flag the generic kernel
// reads at its start -- both are enqueued, the second is a no-op when the first one succeeded.
// 16 384 < N <= 1 966 080: the cooperative form (85-120 us; one workgroup with 64 keys per thread
// took 114-134 us up to 65 536 and 0.4-4.8 ms streaming beyond).  Larger: one workgroup, streaming.
__global__ void threshold_flag_clear(int32_t *flag) { *flag = 0; }

template <bool TRUNC>
static int launch_threshold(float *w, int64_t N, float alpha, float *thr, uint8_t *mask,
                            int64_t *kept, void *ws, hipStream_t st) {
    int32_t *flag = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + WS_SCRATCH_OFF);
    static const int64_t coop_nmin = getenv("RLVI_THR_COOP_NMIN") ? atoll(getenv("RLVI_THR_COOP_NMIN")) : (int64_t)THR_BLOCK * 16 + 1;
    if (N <= (int64_t)THR_BLOCK * 16 && N < coop_nmin) {
        hipLaunchKernelGGL(threshold_flag_clear, dim3(1), dim3(1), 0, st, flag);
        hipLaunchKernelGGL((threshold_fast_kernel<16, TRUNC>), dim3(1), dim3(THR_BLOCK), 0, st, w, N,
                           alpha, thr, mask, kept, flag);
        hipLaunchKernelGGL((threshold_kernel<0, TRUNC>), dim3(1), dim3(THR_BLOCK), 0, st, w, N, alpha,
                           thr, mask, kept, flag);
    } else if (N <= (int64_t)THC_G * THC_BLOCK * 32) {
        const int64_t L = (N + THC_G - 1) / THC_G;
#define RLVI_THC(E_)                                                                             \
    hipLaunchKernelGGL((threshold_coop_kernel<E_, TRUNC>), dim3(THC_G), dim3(THC_BLOCK), 0, st, w, \
                       N, alpha, thr, mask, kept, ws)
        if (L <= THC_BLOCK * 2) RLVI_THC(2);
        else if (L <= THC_BLOCK * 8) RLVI_THC(8);
        else RLVI_THC(32);
#undef RLVI_THC
    } else {
        hipLaunchKernelGGL((threshold_kernel<0, TRUNC>), dim3(1), dim3(THR_BLOCK), 0, st, w, N, alpha,
                           thr, mask, kept, (const int32_t *)nullptr);
    }
    return (int)hipGetLastError();
}

}  // namespace rlvi

using namespace rlvi;

extern "C" int rlvi_fn_threshold_f32(const float *weights, int64_t N, float alpha, float *thr_out,
                                     void *ws, void *stream) {
    if (!weights || !thr_out || !ws) return RLVI_E_NULL;
    if (N <= 0) return RLVI_E_SHAPE;
    if (N > (1ll << 29)) return RLVI_E_LIMIT;   // exactness bound of the fp64 prefix sums
    if (((uintptr_t)weights & 3) || ((uintptr_t)thr_out & 3)) return RLVI_E_ALIGN;
    return launch_threshold<false>(const_cast<float *>(weights), N, alpha, thr_out, nullptr,
                                   nullptr, ws, static_cast<hipStream_t>(stream));
}

extern "C" int rlvi_threshold_truncate_f32(float *weights, int64_t N, float alpha,
                                           float *thr_inout, uint8_t *mask_gt, int64_t *kept_out,
                                           void *ws, void *stream) {
    if (!weights || !thr_inout || !ws) return RLVI_E_NULL;
    if (N <= 0) return RLVI_E_SHAPE;
    if (N > (1ll << 29)) return RLVI_E_LIMIT;
    if (((uintptr_t)weights & 3) || ((uintptr_t)thr_inout & 3) || ((uintptr_t)kept_out & 7))
        return RLVI_E_ALIGN;
    return launch_threshold<true>(weights, N, alpha, thr_inout, mask_gt, kept_out, ws,
                                  static_cast<hipStream_t>(stream));
}

extern "C" int rlvi_truncate_f32(float *weights, int64_t N, const float *thr, uint8_t *mask_gt,
                                 void *stream) {
    if (!weights || !thr) return RLVI_E_NULL;
    if (N <= 0) return RLVI_E_SHAPE;
    int64_t nb = (N + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(truncate_kernel, dim3((unsigned)nb), dim3(256), 0,
                       static_cast<hipStream_t>(stream), weights, N, thr, mask_gt);
    return (int)hipGetLastError();
}
