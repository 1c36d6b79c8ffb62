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
#include "rlvi_common.h"

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
                                                              int64_t *__restrict__ kept_out) {
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

template <bool TRUNC>
static int launch_threshold(float *w, int64_t N, float alpha, float *thr, uint8_t *mask,
                            int64_t *kept, hipStream_t st) {
    if (N <= (int64_t)THR_BLOCK * 16)
        hipLaunchKernelGGL((threshold_kernel<16, TRUNC>), dim3(1), dim3(THR_BLOCK), 0, st, w, N,
                           alpha, thr, mask, kept);
    else if (N <= (int64_t)THR_BLOCK * 80)
        hipLaunchKernelGGL((threshold_kernel<80, TRUNC>), dim3(1), dim3(THR_BLOCK), 0, st, w, N,
                           alpha, thr, mask, kept);
    else
        hipLaunchKernelGGL((threshold_kernel<0, TRUNC>), dim3(1), dim3(THR_BLOCK), 0, st, w, N,
                           alpha, thr, mask, kept);
    return (int)hipGetLastError();
}

}  // namespace rlvi

using namespace rlvi;

extern "C" int rlvi_fn_threshold_f32(const float *weights, int64_t N, float alpha, float *thr_out,
                                     void *ws, void *stream) {
    (void)ws;
    if (!weights || !thr_out) return RLVI_E_NULL;
    if (N <= 0) return RLVI_E_SHAPE;
    if (N > (1ll << 29)) return RLVI_E_LIMIT;   // exactness bound of the fp64 prefix sums
    if (((uintptr_t)weights & 3) || ((uintptr_t)thr_out & 3)) return RLVI_E_ALIGN;
    return launch_threshold<false>(const_cast<float *>(weights), N, alpha, thr_out, nullptr,
                                   nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int rlvi_threshold_truncate_f32(float *weights, int64_t N, float alpha,
                                           float *thr_inout, uint8_t *mask_gt, int64_t *kept_out,
                                           void *ws, void *stream) {
    (void)ws;
    if (!weights || !thr_inout) return RLVI_E_NULL;
    if (N <= 0) return RLVI_E_SHAPE;
    if (N > (1ll << 29)) return RLVI_E_LIMIT;
    if (((uintptr_t)weights & 3) || ((uintptr_t)thr_inout & 3) || ((uintptr_t)kept_out & 7))
        return RLVI_E_ALIGN;
    return launch_threshold<true>(weights, N, alpha, thr_inout, mask_gt, kept_out,
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
