// M-step over one mini-batch: per-sample NLL + top-1 + residual scatter + pi gather +
// pi-weighted loss + gradient w.r.t. the logits, in ONE streaming pass over the logit block.
//
// Replaces deep-learning/methods/train_rlvi.py:85,89,90,92,93-94 and the autograd backward
// reached from :96.
//
// Bound: HBM.  Algorithmic bytes per sample = 2*C*s + 24 (logits read + grad write + label 8 +
// index 8 + pi gather 4 + residual scatter 4), s = 4 (fp32) or 2 (bf16).
//
// Mapping (gfx950, 64-lane waves): a row is owned by a group of G consecutive lanes, every lane
// holding K vectors of V elements (16 B per lane per load when the row pitch allows), so a wave
// works on 64/G rows at once and a row is read exactly once and written exactly once.  Row
// reductions (max, sum-exp, arg-max) are butterfly shuffles inside the lane group.
#include "rlvi_common.h"

namespace rlvi {

template <typename T, int V>
struct VecIO;

template <>
struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct VecIO<float, 2> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[2]) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    }
};
template <>
struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct VecIO<uint16_t, 8> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <>
struct VecIO<uint16_t, 4> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[4]) {
        uint2 t;
        t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2 *>(p) = t;
    }
};
template <>
struct VecIO<uint16_t, 2> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[2]) {
        const uint32_t t = *reinterpret_cast<const uint32_t *>(p);
        v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[2]) {
        *reinterpret_cast<uint32_t *>(p) =
            (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    }
};
template <>
struct VecIO<uint16_t, 1> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[1]) {
        v[0] = bf16_to_f32(*p);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[1]) {
        *p = f32_to_bf16(v[0]);
    }
};

constexpr int MSTEP_THREADS = 256;
constexpr int MSTEP_WAVES = MSTEP_THREADS / WAVE;

template <typename T, int V, int G, int K>
__global__ __launch_bounds__(MSTEP_THREADS) void mstep_kernel(
    const T *__restrict__ logits, int64_t ld, const int64_t *__restrict__ labels,
    const int64_t *__restrict__ idx, const float *__restrict__ weights,
    float *__restrict__ residuals, int64_t N, int64_t B, int C, float inv_scale,
    T *__restrict__ grad, int64_t ldg, double *__restrict__ part, int32_t *__restrict__ status) {
    constexpr int R = WAVE / G;  // rows per wave
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const float NEG_INF = -__builtin_inff();

    float acc = 0.0f;   // sum of pi*l over the rows whose lane-group leader this lane is
    float hits = 0.0f;
    bool bad = false;

    const int64_t stride = (int64_t)gridDim.x * MSTEP_WAVES * R;
    for (int64_t row0 = ((int64_t)blockIdx.x * MSTEP_WAVES + wave) * R; row0 < B; row0 += stride) {
        const int64_t row = row0 + sub;
        const bool valid = row < B;
        const int64_t rr = valid ? row : B - 1;   // padding rows recompute the last row, store nothing
        const T *zrow = logits + rr * ld;

        float v[K][V];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int col = (k * G + g) * V;
            if (col < C) {
                VecIO<T, V>::load(zrow + col, v[k]);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) v[k][j] = NEG_INF;
            }
        }
        int64_t y64 = labels[rr];
        int64_t ix = idx != nullptr ? idx[rr] : rr;   // idx == NULL: identity (in-batch E+M)
        bool row_ok = valid;
        if (y64 < 0 || y64 >= C) { y64 = 0; bad = bad || valid; row_ok = false; }
        if (ix < 0 || ix >= N) { ix = 0; bad = bad || valid; row_ok = false; }
        const int y = (int)y64;
        const float pi = weights[ix];

        float m = NEG_INF;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) m = fmaxf(m, v[k][j]);
        m = group_max<G>(m);

        float s = 0.0f;
        int first = 0x7FFFFFFF;
        float cand = 0.0f;
        const int yv = y / V, jy = y - yv * V;
        const int ky = yv / G, gy = yv & (G - 1);
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
#pragma unroll
            for (int j = V - 1; j >= 0; --j) {
                const float z = v[k][j];
                if (z == m) first = (k * G + g) * V + j;   // descending scan keeps the lowest column
                if (k == ky && j == jy) cand = z;
                const float e = expf(z - m);
                v[k][j] = e;
                s += e;
            }
        }
        s = group_sum<G>(s);
        const int amax = group_min_i<G>(first);
        const float zy = __shfl(cand, sub * G + gy, WAVE);
        const float logs = logf(s);
        const float li = logs - (zy - m);   // == -((z_y - max) - log(sum exp)), as torch evaluates it

        if (grad != nullptr && row_ok) {
            const float gs = pi * inv_scale;
            const float inv_s = gs / s;
            T *grow = grad + rr * ldg;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int col = (k * G + g) * V;
                if (col < C) {
                    float o[V];
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        float p = v[k][j] * inv_s;
                        if (col + j == y) p -= gs;
                        o[j] = p;
                    }
                    VecIO<T, V>::store(grow + col, o);
                }
            }
        }
        if (g == 0 && row_ok) {
            if (residuals != nullptr) residuals[ix] = li;
            acc += li * pi;
            hits += (amax == y) ? 1.0f : 0.0f;
        }
    }

    // block partials -> workspace (fixed order: deterministic)
    double a = wave_sum((double)acc);
    double h = wave_sum((double)hits);
    __shared__ double sh[2 * MSTEP_WAVES];
    if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = h; }
    if (bad) atomicOr(status, RLVI_ST_RANGE);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, th = 0.0;
#pragma unroll
        for (int w = 0; w < MSTEP_WAVES; ++w) { ta += sh[2 * w]; th += sh[2 * w + 1]; }
        part[2 * blockIdx.x] = ta;
        part[2 * blockIdx.x + 1] = th;
    }
}

// Sums the per-block partials in a fixed order and writes the four output scalars.
__global__ __launch_bounds__(256) void mstep_finalize_kernel(const double *__restrict__ part,
                                                             int nblocks, float inv_scale,
                                                             double inv_rows100,
                                                             float *__restrict__ out) {
    double a = 0.0, h = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += part[2 * i]; h += part[2 * i + 1]; }
    a = wave_sum(a);
    h = wave_sum(h);
    __shared__ double sh[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, th = 0.0;
        for (int w = 0; w < 4; ++w) { ta += sh[2 * w]; th += sh[2 * w + 1]; }
        out[0] = (float)(ta * (double)inv_scale);
        out[1] = (float)th * (float)inv_rows100;
        out[2] = (float)ta;
        out[3] = (float)th;
    }
}

template <typename T, int V, int G, int K>
static int launch_mstep(const T *logits, int64_t ld, const int64_t *labels, const int64_t *idx,
                        const float *weights, float *residuals, int64_t N, int64_t B, int C,
                        float inv_scale, T *grad, int64_t ldg, float *out, void *ws,
                        hipStream_t st) {
    constexpr int R = WAVE / G;
    const int64_t rows_per_block = (int64_t)MSTEP_WAVES * R;
    int64_t nb = (B + rows_per_block - 1) / rows_per_block;
    if (nb > MSTEP_MAX_BLOCKS) nb = MSTEP_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    char *base = static_cast<char *>(ws);
    double *part = reinterpret_cast<double *>(base + WS_PART_OFF);
    int32_t *status = reinterpret_cast<int32_t *>(base);
    hipLaunchKernelGGL((mstep_kernel<T, V, G, K>), dim3((unsigned)nb), dim3(MSTEP_THREADS), 0, st,
                       logits, ld, labels, idx, weights, residuals, N, B, C, inv_scale, grad, ldg,
                       part, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mstep_finalize_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, inv_scale,
                       100.0 / (double)B, out);
    e = hipGetLastError();
    return (int)e;
}

// Picks (V, G, K): V from the alignment of base pointers / pitches / C, then the smallest lane
// group that covers ceil(C/V) vectors with at most 8 vectors per lane.
template <typename T, int V>
static int dispatch_gk(const T *logits, int64_t ld, const int64_t *labels, const int64_t *idx,
                       const float *weights, float *residuals, int64_t N, int64_t B, int C,
                       float inv_scale, T *grad, int64_t ldg, float *out, void *ws,
                       hipStream_t st) {
    const int nv = (C + V - 1) / V;
#define RLVI_CASE(G_, K_)                                                                        \
    return launch_mstep<T, V, G_, K_>(logits, ld, labels, idx, weights, residuals, N, B, C,      \
                                      inv_scale, grad, ldg, out, ws, st)
    if (nv <= 1) RLVI_CASE(1, 1);
    if (nv <= 2) RLVI_CASE(2, 1);
    if (nv <= 4) RLVI_CASE(4, 1);
    if (nv <= 8) RLVI_CASE(8, 1);
    if (nv <= 16) RLVI_CASE(16, 1);
    if (nv <= 32) RLVI_CASE(32, 1);
    if (nv <= 64) RLVI_CASE(64, 1);
    if (nv <= 128) RLVI_CASE(64, 2);
    if (nv <= 256) RLVI_CASE(64, 4);
    if (nv <= 512) RLVI_CASE(64, 8);
#undef RLVI_CASE
    return RLVI_E_LIMIT;
}

template <typename T>
static int mstep_entry(const T *logits, int64_t ld, const int64_t *labels, const int64_t *idx,
                       const float *weights, float *residuals, int64_t N, int64_t B, int64_t C,
                       float inv_scale, T *grad, int64_t ldg, float *out, void *ws, void *stream) {
    if (!logits || !labels || !weights || !out || !ws) return RLVI_E_NULL;
    if (B <= 0 || C <= 0 || N <= 0 || ld < C || (grad && ldg < C)) return RLVI_E_SHAPE;
    if (C > (1 << 20)) return RLVI_E_LIMIT;
    if (((uintptr_t)labels & 7) || ((uintptr_t)idx & 7) || ((uintptr_t)weights & 3) ||
        ((uintptr_t)residuals & 3) || ((uintptr_t)out & 3) || ((uintptr_t)ws & 255) ||
        ((uintptr_t)logits % sizeof(T)) || (grad && ((uintptr_t)grad % sizeof(T))))
        return RLVI_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // widest vector for which every row start and the row length are aligned
    auto ok = [&](int v) {
        const size_t bytes = (size_t)v * sizeof(T);
        if (C % v || ld % v || ((uintptr_t)logits % bytes)) return false;
        if (grad && (ldg % v || ((uintptr_t)grad % bytes))) return false;
        return true;
    };
    constexpr int VMAX = 16 / (int)sizeof(T);
    const int Ci = (int)C;
    if constexpr (VMAX == 8) {
        if (ok(8))
            return dispatch_gk<T, 8>(logits, ld, labels, idx, weights, residuals, N, B, Ci,
                                     inv_scale, grad, ldg, out, ws, st);
    }
    if (ok(4))
        return dispatch_gk<T, 4>(logits, ld, labels, idx, weights, residuals, N, B, Ci, inv_scale,
                                 grad, ldg, out, ws, st);
    if (ok(2))
        return dispatch_gk<T, 2>(logits, ld, labels, idx, weights, residuals, N, B, Ci, inv_scale,
                                 grad, ldg, out, ws, st);
    return dispatch_gk<T, 1>(logits, ld, labels, idx, weights, residuals, N, B, Ci, inv_scale,
                             grad, ldg, out, ws, st);
}

}  // namespace rlvi

extern "C" int rlvi_mstep_fwd_bwd_f32(const float *logits, int64_t ld, const int64_t *labels,
                                      const int64_t *idx, const float *weights, float *residuals,
                                      int64_t N, int64_t B, int64_t C, float inv_scale,
                                      float *grad_logits, int64_t ldg, float *out, void *ws,
                                      void *stream) {
    return rlvi::mstep_entry<float>(logits, ld, labels, idx, weights, residuals, N, B, C,
                                    inv_scale, grad_logits, ldg, out, ws, stream);
}

extern "C" int rlvi_mstep_fwd_bwd_bf16(const uint16_t *logits, int64_t ld, const int64_t *labels,
                                       const int64_t *idx, const float *weights, float *residuals,
                                       int64_t N, int64_t B, int64_t C, float inv_scale,
                                       uint16_t *grad_logits, int64_t ldg, float *out, void *ws,
                                       void *stream) {
    return rlvi::mstep_entry<uint16_t>(logits, ld, labels, idx, weights, residuals, N, B, C,
                                       inv_scale, grad_logits, ldg, out, ws, stream);
}
