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
#include <stdlib.h>

#include <type_traits>

#include "rlvi_common.h"

namespace rlvi {

#ifndef RLVI_MSTEP_NT
#define RLVI_MSTEP_NT 0   // bit 0: nontemporal row loads, bit 1: nontemporal gradient stores
#endif
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
typedef unsigned int vu2 __attribute__((ext_vector_type(2)));

template <typename T, int V>
struct VecIO;

// default streaming forms = the plain forms (specialisations below override where it pays)
template <typename T, int V, class Self>
struct VecIOBase {
    static __device__ __forceinline__ void load_stream(const T *p, float (&v)[V]) { Self::load(p, v); }
    static __device__ __forceinline__ void store_stream(T *p, const float (&v)[V]) { Self::store(p, v); }
};

template <>
struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // streaming forms: read-once / write-once data bypasses cache retention (`nt`)
    static __device__ __forceinline__ void load_stream(const float *p, float (&v)[4]) {
#if RLVI_MSTEP_NT & 1
        const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
#else
        load(p, v);
#endif
    }
    static __device__ __forceinline__ void store_stream(float *p, const float (&v)[4]) {
#if RLVI_MSTEP_NT & 2
        const vf4 t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(p));
#else
        store(p, v);
#endif
    }
};
template <>
struct VecIO<float, 2> : VecIOBase<float, 2, VecIO<float, 2>> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[2]) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    }
};
template <>
struct VecIO<float, 1> : VecIOBase<float, 1, VecIO<float, 1>> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct VecIO<uint16_t, 8> : VecIOBase<uint16_t, 8, VecIO<uint16_t, 8>> {
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
            w[i] = f32x2_to_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <>
struct VecIO<uint16_t, 4> : VecIOBase<uint16_t, 4, VecIO<uint16_t, 4>> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[4]) {
        uint2 t;
        t.x = f32x2_to_bf16x2(v[0], v[1]);
        t.y = f32x2_to_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p) = t;
    }
};
template <>
struct VecIO<uint16_t, 2> : VecIOBase<uint16_t, 2, VecIO<uint16_t, 2>> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[2]) {
        const uint32_t t = *reinterpret_cast<const uint32_t *>(p);
        v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[2]) {
        *reinterpret_cast<uint32_t *>(p) = f32x2_to_bf16x2(v[0], v[1]);
    }
};
template <>
struct VecIO<uint16_t, 1> : VecIOBase<uint16_t, 1, VecIO<uint16_t, 1>> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[1]) {
        v[0] = bf16_to_f32(*p);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[1]) {
        *p = f32_to_bf16(v[0]);
    }
};

constexpr int MSTEP_THREADS = 256;
constexpr int MSTEP_WAVES = MSTEP_THREADS / WAVE;

// Per-block partial record.  accum == 0: overwrite (a finalize launch follows); accum == 1: add
// to what earlier mini-batches of this epoch left there (rlvi_epoch_end_f32 reduces and clears).
// Every block owns its record, so there are no atomics and the sums are order-deterministic.
__device__ __forceinline__ void write_partial(double *part, double ta, double th, float inv_scale,
                                              double inv_rows100, int accum) {
    double *p = part + (size_t)PART_STRIDE * blockIdx.x;
    const double v0 = ta * (double)inv_scale, v1 = th * inv_rows100;
#ifndef RLVI_MSTEP_ATOMIC_TAIL
#define RLVI_MSTEP_ATOMIC_TAIL 1
#endif
    if (accum && !RLVI_MSTEP_ATOMIC_TAIL) { p[0] += v0; p[1] += v1; p[2] += ta; p[3] += th; }
    else if (accum) {
        // four no-return fp64 adds at the memory side instead of a read-modify-write: the record's old
        // value never travels, so the last workgroups of a launch end with four posted operations and not
        // with a load round trip (~1 us) on the launch's critical path.  One adder per record and launch,
        // launches in stream order: the sums stay order-deterministic.
        typedef __attribute__((address_space(1))) double gdouble;
        gdouble *q = (gdouble *)p;
        __builtin_amdgcn_global_atomic_fadd_f64(q + 0, v0);
        __builtin_amdgcn_global_atomic_fadd_f64(q + 1, v1);
        __builtin_amdgcn_global_atomic_fadd_f64(q + 2, ta);
        __builtin_amdgcn_global_atomic_fadd_f64(q + 3, th);
    } else { p[0] = v0; p[1] = v1; p[2] = ta; p[3] = th; }
}

// A row is owned by G consecutive lanes; lane g holds the vectors k*G+g, k < kact <= KMAX, so one
// load instruction reads G*V contiguous elements of each of the wave's 64/G rows and a lane
// amortises the (short) lane-group reductions over up to KMAX*V elements.
template <typename T, int V, int G, int KMAX>
__global__ __launch_bounds__(MSTEP_THREADS) void mstep_kernel(
    const T *__restrict__ logits, int64_t ld, const int64_t *__restrict__ labels,
    const int64_t *__restrict__ idx, const float *__restrict__ weights,
    float *__restrict__ residuals, int64_t N, int64_t B, int C, int kact, float inv_scale,
    T *__restrict__ grad, int64_t ldg, double *__restrict__ part, int32_t *__restrict__ status,
    int accum, double inv_rows100, int64_t first_row) {
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

        // Issue order matters: the small label / index reads go FIRST so that they return ahead
        // of the row data (loads of a wave return in order) and the dependent pi gather and
        // label-logit read are already queued while the row vectors are still in flight --
        // otherwise they cost a second full memory round trip behind everybody's row data.
        int64_t y64 = labels[rr];
        int64_t ix = (idx != nullptr ? idx : labels)[rr];   // unconditional load, selected below

        // branch-free row loads: a lane's unused vector slots re-read its first vector (always in
        // range, same cache line) and are masked to -inf afterwards, so the compiler can count
        // the outstanding loads instead of draining them at a branch
        float v[KMAX][V];
        bool live[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int col = (k * G + g) * V;
            live[k] = k < kact && col < C;
            VecIO<T, V>::load_stream(zrow + (live[k] ? col : g * V), v[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
        ix = idx != nullptr ? ix : first_row + rr;    // idx == NULL: identity (in-batch E+M), counted from the batch start
        bool row_ok = valid;
        if (y64 < 0 || y64 >= C) { y64 = 0; bad = bad || valid; row_ok = false; }
        if (ix < 0 || ix >= N) { ix = 0; bad = bad || valid; row_ok = false; }
        const int y = (int)y64;
        const float pi = weights != nullptr ? weights[ix] : 1.0f;   // no weights: plain CE
        // the label logit again (one 4-byte read of a line this wave has just requested)
        float zy;
        {
            float t[1];
            VecIO<T, 1>::load(zrow + y, t);
            zy = t[0];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the two dependent reads queued behind the row data

        float m = NEG_INF;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                v[k][j] = live[k] ? v[k][j] : NEG_INF;
                m = fmaxf(m, v[k][j]);
            }
        m = group_max<G>(m);

        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float e = mexp(v[k][j] - m);
                v[k][j] = e;
                s += e;
            }
        }
        s = group_sum<G>(s);
        const float logs = logf(s);
        const float li = logs - (zy - m);   // == -((z_y - max) - log(sum exp)), as torch evaluates it

        if (grad != nullptr && valid) {
            const float gs = row_ok ? pi * inv_scale : 0.0f;   // a rejected row gets a zero gradient row
            const float inv_s = gs / s;
            T *grow = grad + rr * ldg;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int col = (k * G + g) * V;
                if (live[k]) {
                    float o[V];
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        float p = v[k][j] * inv_s;
                        if (col + j == y) p -= gs;
                        o[j] = p;
                    }
                    VecIO<T, V>::store_stream(grow + col, o);
                }
            }
        }
        // top-1: the label is a hit when it is the FIRST column that attains the row maximum
        // (torch.max order, deep-learning/utils.py:58); two exact maxima put at least 2.0 into the
        // sum, so the exact check (a second read of the row) runs only for waves that hold such a row
        bool hit = zy == m;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit && s >= 2.0f) != 0, 0)) {
            int earlier = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int col = (k * G + g) * V;
                float z[V];
                VecIO<T, V>::load(zrow + (live[k] ? col : g * V), z);
#pragma unroll
                for (int j = 0; j < V; ++j) earlier += (live[k] && z[j] == m && col + j < y) ? 1 : 0;
            }
            earlier = group_allreduce<G>(earlier, FAdd());
            hit = hit && earlier == 0;
        }
        if (g == 0 && row_ok) {
            if (residuals != nullptr) residuals[ix] = li;
            acc += li * pi;
            hits += hit ? 1.0f : 0.0f;
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
        write_partial(part, ta, th, inv_scale, inv_rows100, accum);
    }
}

// ---------------------------------------------------------------------------------------
// Long rows (more than 512 vectors: C > 2048 fp32 / 4096 bf16 aligned, C > 512 in single elements -- an odd
// ImageNet-21k head, a padded pitch): a row does not fit a wave's registers, so ONE WAVE walks its row three times
// from global memory (the second and third pass are L2 hits): maximum -- sum of exponentials, label logit and the
// ties in front of the label -- gradient.  Same arithmetic and the same records as the register-row kernel.
// WPR = waves per row: 1 (a wave per row: many rows) or MSTEP_WAVES (the whole workgroup on one row: batches of
// fewer rows than the chip has wave slots -- 256 x 21 841 is 256 rows).  Four vectors per lane in flight per trip.
template <typename T, int V, int WPR>
__global__ __launch_bounds__(MSTEP_THREADS) void mstep_longrow_kernel(
    const T *__restrict__ logits, int64_t ld, const int64_t *__restrict__ labels,
    const int64_t *__restrict__ idx, const float *__restrict__ weights,
    float *__restrict__ residuals, int64_t N, int64_t B, int C, float inv_scale,
    T *__restrict__ grad, int64_t ldg, double *__restrict__ part, int32_t *__restrict__ status,
    int accum, double inv_rows100) {
    constexpr int U = 4;                                          // vectors per lane and trip
    constexpr int RPB = MSTEP_WAVES / WPR;                        // rows per workgroup at a time
    constexpr int STEP = WPR * WAVE;                              // lanes on one row
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int t = WPR == 1 ? lane : threadIdx.x;                  // this thread's position on its row
    __shared__ float shf[2][MSTEP_WAVES];
    __shared__ int shi[MSTEP_WAVES];
    float acc = 0.0f, hits = 0.0f;
    bool bad = false;
    const int nv = C / V;                                         // (V divides C)
    const int64_t stride = (int64_t)gridDim.x * RPB;
    // (every thread of a workgroup runs the same number of trips: the barriers below are uniform)
    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < B; row0 += stride) {
        const int64_t row = row0 + (WPR == 1 ? wave : 0);
        const bool valid = row < B;
        const int64_t rr = valid ? row : B - 1;                   // a wave without a row recomputes the last one, stores nothing
        const T *zrow = logits + rr * ld;
        int64_t y64 = labels[rr];
        int64_t ix = idx != nullptr ? idx[rr] : rr;
        bool row_ok = valid;
        if (y64 < 0 || y64 >= C) { y64 = 0; bad = bad || valid; row_ok = false; }
        if (ix < 0 || ix >= N) { ix = 0; bad = bad || valid; row_ok = false; }
        const int y = (int)y64;
        const float pi = weights != nullptr ? weights[ix] : 1.0f;
        float zy;
        {
            float tt[1];
            VecIO<T, 1>::load(zrow + y, tt);
            zy = tt[0];
        }
        // ---- pass 1: the row maximum
        float m = -__builtin_inff();
        for (int k0 = t; k0 < nv; k0 += STEP * U) {
            float v[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * STEP;
                VecIO<T, V>::load(zrow + (size_t)(k < nv ? k : k0) * V, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) m = fmaxf(m, v[u][j]);      // (a slot past the end repeats k0: harmless for a maximum)
        }
        m = group_max<WAVE>(m);
        if (WPR > 1) {
            if (lane == 0) shf[0][wave] = m;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < MSTEP_WAVES; ++w) m = fmaxf(m, shf[0][w]);
        }
        // ---- pass 2: the sum of exponentials and the ties in front of the label
        float s = 0.0f;
        int earlier = 0;
        for (int k0 = t; k0 < nv; k0 += STEP * U) {
            float v[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * STEP;
                VecIO<T, V>::load(zrow + (size_t)(k < nv ? k : k0) * V, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * STEP;
                const bool in = k < nv;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    s += in ? mexp(v[u][j] - m) : 0.0f;
                    earlier += (in && v[u][j] == m && k * V + j < y) ? 1 : 0;
                }
            }
        }
        s = group_sum<WAVE>(s);
        earlier = group_allreduce<WAVE>(earlier, FAdd());
        if (WPR > 1) {
            if (lane == 0) { shf[1][wave] = s; shi[wave] = earlier; }
            __syncthreads();
            s = 0.0f;
            earlier = 0;
#pragma unroll
            for (int w = 0; w < MSTEP_WAVES; ++w) { s += shf[1][w]; earlier += shi[w]; }
        }
        const float li = logf(s) - (zy - m);
        // ---- pass 3: the gradient
        if (grad != nullptr && valid) {
            const float gs = row_ok ? pi * inv_scale : 0.0f;
            const float inv_s = gs / s;
            T *grow = grad + rr * ldg;
            for (int k0 = t; k0 < nv; k0 += STEP * U) {
                float v[U][V];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u * STEP;
                    VecIO<T, V>::load(zrow + (size_t)(k < nv ? k : k0) * V, v[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u * STEP;
                    if (k < nv) {
                        float o[V];
#pragma unroll
                        for (int j = 0; j < V; ++j) {
                            float p = mexp(v[u][j] - m) * inv_s;
                            if (k * V + j == y) p -= gs;
                            o[j] = p;
                        }
                        VecIO<T, V>::store_stream(grow + (size_t)k * V, o);
                    }
                }
            }
        }
        const bool hit = zy == m && earlier == 0;
        if (t == 0 && row_ok) {
            if (residuals != nullptr) residuals[ix] = li;
            acc += li * pi;
            hits += hit ? 1.0f : 0.0f;
        }
        if (WPR > 1) __syncthreads();                             // (shf / shi are rewritten by the next row)
    }
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
        write_partial(part, ta, th, inv_scale, inv_rows100, accum);
    }
}

// ---------------------------------------------------------------------------------------
// Wave-tile form (dense rows): every WAVE streams its own tiles of R = 64/G rows through its own
// slice of LDS -- no workgroup barrier anywhere, and the hot loop is straight-line code:
//   A  label / index (asm loads the compiler does not count), then the tile global -> LDS by
//      LDS-DMA (global_load_lds_dwordx4: flat, 16 B per lane, 1 KiB per wave instruction, no
//      staging registers); the wait for "all but the DMA pieces" hands over label / index while
//      the tile is still in flight, so the dependent pi gather flies beside the tile
//   B  G-lane groups own the rows as in the other forms.  A lane's slots past the end of the row
//      alias the row's LAST vector: they read what its owner reads (harmless for the maximum),
//      are masked out of the sum, and their in-place gradient write repeats the owner's value --
//      no execution-mask juggling.  The -onehot term is one LDS read-modify-write per row (all
//      lanes of the group write the same value)
//   C  flat nontemporal 16-B/lane stores LDS -> grad
// Only full tiles; the launcher hands the B mod R trailing rows to the register-row kernel.
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
// RLVI_MSTEP_DMA=1: stage the tile by LDS-DMA (global_load_lds_dwordx4) instead of nontemporal
// register loads + ds_write.  Measured slower at 65 536 x 100 (12.3 against 10.4 us): a CU accepts
// DMA pieces only about as fast as they return, so the last workgroup of a CU issues its reads 3 us
// after the first, while register loads of all 16 waves of a CU are in flight at once.
#ifndef RLVI_MSTEP_DMA
#define RLVI_MSTEP_DMA 0
#endif
#ifndef RLVI_MSTEP_EARLY_PI
#define RLVI_MSTEP_EARLY_PI 0
#endif
#ifndef RLVI_MSTEP_DMA_AUX
#define RLVI_MSTEP_DMA_AUX 2      // cache policy of the tile DMA: 2 = nt (read once)
#endif
// -DRLVI_MSTEP_STAMPS: diagnostic build, every wave leaves wall-clock stamps (100 MHz) of its first
// tile's phases in the workspace scratch (tools/mstep_stamps.py); never in the product library.
#ifdef RLVI_MSTEP_STAMPS
#define RLVI_STAMP(k) do { if (lane == 0 && first_tile) stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RLVI_STAMP(k) do { } while (0)
#endif
#ifndef RLVI_MSTEP_WAVE_MINW
#define RLVI_MSTEP_WAVE_MINW 4    // waves per SIMD the register allocation must allow (LDS allows 5)
#endif

struct FMaxF { __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fmaxf(a, b); } };

// EXACT: the host guarantees ceil(ceil(C/V)/G) == KMAX, so only a lane's LAST slot can lie past
// the end of the row and (for 16-byte vectors) only the last DMA / store piece can be partial.
template <typename T, int V, int G, int KMAX, int WPB, bool EXACT>
__global__ __launch_bounds__(WPB *WAVE, RLVI_MSTEP_WAVE_MINW) void mstep_wave_kernel(
    const T *__restrict__ logits, const int64_t *__restrict__ labels,
    const int64_t *__restrict__ idx, const float *__restrict__ weights,
    float *__restrict__ residuals, int64_t N, int64_t nfull, int C, float inv_scale,
    T *__restrict__ grad, double *__restrict__ part, int32_t *__restrict__ status, int accum,
    double inv_rows100, int hold_ticks, int gen_ticks, unsigned long long *__restrict__ hold_slot,
    unsigned long long hold_key, int hold_cap, int hold_pct) {
    constexpr int R = WAVE / G;                                   // rows per wave tile
    constexpr int VB = V * (int)sizeof(T);                        // bytes of a lane vector
    constexpr int NI = (KMAX * VB + 15) / 16;                     // 1-KiB pieces per tile (max)
    constexpr int WTILE = NI * 1024;                              // LDS bytes per wave
    constexpr bool EXACT16 = EXACT && VB == 16;                   // then pieces 0..NI-2 are full
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const int nv = C / V;                                         // vectors per row (V divides C)
    const int nchunk = (R * C * (int)sizeof(T)) >> 4;            // 16-byte chunks of a tile
    char *wtile = smem + (size_t)wave * WTILE;
    vu4 *tile16 = reinterpret_cast<vu4 *>(wtile);
    const int64_t tstride = (int64_t)gridDim.x * WPB;
#ifdef RLVI_MSTEP_STAMPS
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(
        reinterpret_cast<char *>(status) + WS_SCRATCH_OFF) + ((size_t)blockIdx.x * WPB + wave) * 16;
    bool first_tile = true;
    RLVI_STAMP(0);
    if (lane == 0) {
        stamps[8] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        stamps[9] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    }
#endif

    // loop-invariant per-lane geometry
    unsigned dma_off[NI];                                         // byte offset of this lane's chunk of piece i
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int c = i * WAVE + lane;
        if (!(EXACT16 && i < NI - 1)) c = c < nchunk ? c : nchunk - 1;   // past the tile: re-read its last chunk
        dma_off[i] = (unsigned)c * 16u;
    }
    int slot_off[KMAX];                                           // LDS byte offset of slot k inside the slice
    bool live[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int vec = k * G + g;
        live[k] = (EXACT && k < KMAX - 1) || vec < nv;
        slot_off[k] = (sub * C + (live[k] ? vec : nv - 1) * V) * (int)sizeof(T);
    }
    const int row_off = sub * C * (int)sizeof(T);
    const int64_t *idxp = idx != nullptr ? idx : labels;
    const unsigned sub8 = (unsigned)sub * 8u;

    float acc = 0.0f, hits = 0.0f;
    bool bad = false;
    // Reads first, then writes -- chip-wide, without a word exchanged: when every wave of the launch has ONE
    // tile (a launch of up to 16 waves x CUs tiles: the bench block), a wave does not issue its gradient
    // stores before `hold_ticks` x 10 ns after ITS OWN start, the time the launch's reads need at the
    // read-only rate of the memory system.  Until then the tile waits in the wave's LDS slice, finished.
    // Stores that start while other waves' tiles are still landing turn the read stream into a mixed
    // read / write stream (5.4 TB/s instead of 6.5 read-only), and the write burst that follows the read
    // phase is absorbed by the Infinity Cache and drains while the next launch is being dispatched:
    // 11.15 -> 10.5 us per launch at 65 536 x 100 (hold 4.0 us; 3.0 and 6.0 us are both slower than none).
    const unsigned long long t_begin = (hold_ticks > 0 || hold_slot != nullptr) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // Self-timed form (hold_slot != nullptr; what a chip-filling one-tile-per-wave launch takes by default): the
    // hold is not an estimate of anybody's -- it is how long the PREVIOUS launch of this shape on this workspace
    // needed to get its tile loads ISSUED, chip-wide (the issue is back-pressured by what the memory system
    // returns, so "the last load is in flight" is when the read phase is as good as over; the per-CU barrier of
    // the 16-wave form waits for exactly that moment, per CU).  A few waves of the launch's last workgroups --
    // the ones a CU starts last -- leave (time bucket << 24 | ticks from their own start to their last load's
    // issue) in the slot with one no-return atomic max: a later launch's stamps supersede an earlier one's, inside
    // a launch the latest bucket's largest value stays.  Logits from the Infinity Cache get their loads issued
    // sooner, so their hold is shorter by itself: no caller hint, no fitted rate.
    unsigned long long slot_key = 0ull, slot_val = 0ull, t_issued = 0ull;
    if (hold_slot != nullptr) { slot_key = hold_slot[0]; slot_val = hold_slot[1]; }

    // CUWIDE (WPB == 16: ONE workgroup per CU, all of its 16 waves): a workgroup barrier right behind the ISSUE
    // of a wave's tile loads -- no wave of the CU sends a store into the CU's memory pipeline before every wave
    // of the CU has its loads in it.  Stamps (tools/mstep_stamps.py) show why that is the moment that counts:
    // the issue of a CU's loads is back-pressured by what the memory system returns -- its last wave issues at
    // 3.6-4.3 us at 65 536 x 100 -- and the first finished waves' stores (from 1.8 us on) queue in front of
    // those loads.  The barrier releases the CU when its last load is in flight: 11.85 -> 10.9 us HBM-cold,
    // 9.5 -> 9.3 us from the Infinity Cache (no estimate of any rate: the data decides), 15.0 -> 13.3 at x 128.
    // (The launcher takes this form only when no wave has a second tile and the grid fills at least four
    //  fifths of the CUs; a wave without a tile only joins the barrier.)
    constexpr bool CUWIDE = WPB == 16;
    if (CUWIDE && (int64_t)blockIdx.x * WPB + wave >= nfull) __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * WPB + wave; t < nfull; t += tstride) {
        const int64_t row_base = t * R;
        // ---- A
        const char *src = reinterpret_cast<const char *>(logits + row_base * C);
        int64_t y64, ix;
        bool okrow = true;
        float pi;
#if RLVI_MSTEP_DMA
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(y64) : "v"(sub8), "s"(labels + row_base) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(ix) : "v"(sub8), "s"(idxp + row_base) : "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t *)(src + dma_off[i]), (lptr_t *)(wtile + i * 1024), 16, 0,
                                             RLVI_MSTEP_DMA_AUX);
        RLVI_STAMP(1);
        // vmcnt(NI): everything older than the NI DMA pieces (= label and index) has returned
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(y64), "+v"(ix) : "n"(NI) : "memory");
        ix = idx != nullptr ? ix : row_base + sub;
        if (y64 < 0 || y64 >= C) { y64 = 0; okrow = false; }
        if (ix < 0 || ix >= N) { ix = 0; okrow = false; }
        pi = weights != nullptr ? weights[ix] : 1.0f;             // flies beside the tile
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pi) : : "memory");   // tile and pi landed
#else
        // label and index first (they return ahead of the tile: loads return in order), the tile's
        // NI chunks per lane behind them
        y64 = *reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(labels + row_base) + sub8);
        ix = *reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(idxp + row_base) + sub8);
        vu4 stg[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            stg[i] = __builtin_nontemporal_load(reinterpret_cast<const vu4 *>(src + dma_off[i]));
        RLVI_STAMP(1);
        if (hold_slot != nullptr) {
            __builtin_amdgcn_sched_barrier(0);       // (the loads are issued, THEN the clock is read)
            t_issued = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (CUWIDE) {
            __builtin_amdgcn_sched_barrier(0);       // (the loads are issued, THEN the barrier)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        ix = idx != nullptr ? ix : row_base + sub;
        if (y64 < 0 || y64 >= C) { y64 = 0; okrow = false; }
        if (ix < 0 || ix >= N) { ix = 0; okrow = false; }
#if RLVI_MSTEP_EARLY_PI
        pi = weights != nullptr ? weights[ix] : 1.0f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) tile16[i * WAVE + lane] = stg[i];
#else
        // the tile first, the gather once it has landed: 64 K random 4-byte reads queued beside the
        // streaming reads cost the launch 1 us (12.0 against 11.0 us at 65 536 x 100); issued here they
        // are L2 hits on a quiet queue and fly during the first half of phase B
#pragma unroll
        for (int i = 0; i < NI; ++i) tile16[i * WAVE + lane] = stg[i];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(y64), "+v"(ix) : : "memory");
        if (hold_slot != nullptr && hold_pct < 0) t_issued = __builtin_amdgcn_s_memrealtime();      // (lab: "landed")
        pi = weights != nullptr ? weights[ix] : 1.0f;
#endif
#endif
        RLVI_STAMP(2);
        bad = bad || !okrow;
        __builtin_amdgcn_wave_barrier();

        // ---- B
        const int y = (int)y64;
        char *zrow = wtile + row_off;
        float zy;
        {
            float tt[1];
            VecIO<T, 1>::load(reinterpret_cast<T *>(zrow) + y, tt);
            zy = tt[0];
        }
        float v[KMAX][V];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) VecIO<T, V>::load(reinterpret_cast<T *>(wtile + slot_off[k]), v[k]);
        float m = v[0][0];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) m = __builtin_fmaxf(m, v[k][j]);
        m = group_allreduce<G>(m, FMaxF());
#ifdef RLVI_MSTEP_STAMPS
        asm volatile("" : "+v"(m));
        RLVI_STAMP(3);
#endif
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float e = mexp(v[k][j] - m);
                v[k][j] = e;
                s += live[k] ? e : 0.0f;
            }
        s = group_sum<G>(s);
#ifdef RLVI_MSTEP_STAMPS
        asm volatile("" : "+v"(s));
        RLVI_STAMP(4);
#endif
        // s is in [1, C]: v_log_f32 needs no range fix-up
        const float li = __builtin_amdgcn_logf(s) * 0.69314718055994530942f - (zy - m);
        // top-1: the label is a hit when it is the FIRST column that attains the row maximum
        // (torch.max order, deep-learning/utils.py:58).  Two exact maxima put at least 2.0 into
        // the sum, so the exact check runs only for waves that hold such a row.
        pi = okrow ? pi : 0.0f;                                   // a rejected row gets a zero gradient
        bool hit = zy == m;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit && s >= 2.0f) != 0, 0)) {
            int earlier = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                float z[V];
                asm volatile("" ::: "memory");                     // re-read, do not keep the first copy live
                VecIO<T, V>::load(reinterpret_cast<T *>(wtile + slot_off[k]), z);
                const int col = (k * G + g) * V;
#pragma unroll
                for (int j = 0; j < V; ++j) earlier += (live[k] && z[j] == m && col + j < y) ? 1 : 0;
            }
            earlier = group_allreduce<G>(earlier, FAdd());
            hit = hit && earlier == 0;
        }
        if (grad != nullptr) {
            const float gs = pi * inv_scale;
            const float inv_s = gs * __builtin_amdgcn_rcpf(s);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    o[j] = v[k][j] * inv_s;
                    if (sizeof(T) != 4) {
                        const int colc = (live[k] ? (k * G + g) : nv - 1) * V + j;
                        if (colc == y) o[j] -= gs;
                    }
                }
                VecIO<T, V>::store(reinterpret_cast<T *>(wtile + slot_off[k]), o);   // in place
            }
            if (sizeof(T) == 4) {
                // -onehot term: one read-modify-write of the label entry, ordered behind this wave's
                // vector stores (LDS operations of a wave complete in order); the lanes of a group
                // all write the same value
                // (gs = pi * inv_scale unrounded inside the fma: spelled out so that fused_em.hip,
                //  which promises the same bits, does not depend on what the compiler contracts)
                float *zf = reinterpret_cast<float *>(zrow);
                zf[y] = fmaf(-inv_scale, pi, zf[y]);
            }
        }
        if (g == 0 && okrow && residuals != nullptr) residuals[ix] = li;
        acc += okrow ? li * pi : 0.0f;
        hits += (hit && okrow) ? 1.0f : 0.0f;
        __builtin_amdgcn_wave_barrier();
        RLVI_STAMP(5);

        // ---- C: flat store of the gradient tile
#ifndef RLVI_MSTEP_NT_STORE
#define RLVI_MSTEP_NT_STORE 1
#endif
#if RLVI_MSTEP_NT_STORE
#define RLVI_TILE_STORE(val, ptr) __builtin_nontemporal_store(val, ptr)
#else
#define RLVI_TILE_STORE(val, ptr) (*(ptr) = (val))
#endif
        if (hold_slot != nullptr) {
            const int pct = hold_pct < 0 ? -hold_pct : hold_pct;
            long long h = slot_key == hold_key ? (long long)(slot_val & 0xFFFFFFull) * pct / 100 : 0ll;
            h = h > hold_cap ? hold_cap : h;
            while ((long long)(__builtin_amdgcn_s_memrealtime() - t_begin) < h) __builtin_amdgcn_s_sleep(4);
            // the reporters: the last wave of every 16th workgroup of the launch's last quarter
            if (lane == 0 && wave == WPB - 1 && (blockIdx.x & 15u) == 15u && blockIdx.x * 4u >= gridDim.x * 3u) {
                unsigned long long dt = t_issued - t_begin;
                dt = dt > 0xFFFFFFull ? 0xFFFFFFull : dt;
                typedef __attribute__((address_space(1))) unsigned long long gull;
                __hip_atomic_store((gull *)hold_slot, hold_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_max((gull *)(hold_slot + 1), ((t_issued >> 6) << 24) | dt, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (hold_ticks > 0) {
            while (__builtin_amdgcn_s_memrealtime() - t_begin < (unsigned long long)hold_ticks)
                __builtin_amdgcn_s_sleep(4);
            hold_ticks = gen_ticks > 0 ? hold_ticks + gen_ticks : 0;      // (lab: one hold per generation of tiles)
        }
        if (grad != nullptr) {
            char *gdst = reinterpret_cast<char *>(grad + row_base * C);
            vu4 st[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) st[i] = tile16[i * WAVE + lane];      // inside this wave's slice
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (EXACT16 && i < NI - 1) {
                    RLVI_TILE_STORE(st[i], reinterpret_cast<vu4 *>(gdst + dma_off[i]));
                } else if ((i + 1) * WAVE <= nchunk) {
                    RLVI_TILE_STORE(st[i], reinterpret_cast<vu4 *>(gdst + dma_off[i]));
                } else if (i * WAVE + lane < nchunk) {
                    RLVI_TILE_STORE(st[i], reinterpret_cast<vu4 *>(gdst + dma_off[i]));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef RLVI_MSTEP_STAMPS
        RLVI_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RLVI_STAMP(7);
        first_tile = false;
#endif
    }

    // every lane of a row's group carried the row's sums: count each row once
    double a = wave_sum((double)(g == 0 ? acc : 0.0f));
    double h = wave_sum((double)(g == 0 ? hits : 0.0f));
    __shared__ double sh[2 * WPB];
    if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = h; }
    if (bad) atomicOr(status, RLVI_ST_RANGE);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, th = 0.0;
#pragma unroll
        for (int w = 0; w < WPB; ++w) { ta += sh[2 * w]; th += sh[2 * w + 1]; }
        write_partial(part, ta, th, inv_scale, inv_rows100, accum);
    }
}

// ---------------------------------------------------------------------------------------
// bf16 rows of an ODD number of elements (C = 101: 202-byte rows, every other row starts in the
// middle of a 32-bit word), at least 32 768 of them (round 4; the round-3 verdict's item 4b).  The general wave tile
// reads such a row one 2-byte element per LDS instruction (27.5 us at 65 536 x 101, so the launcher sent the shape
// to the register-row kernel: 14.3 us).  Here the tile is the same flat 16-B/lane stream into LDS, but a lane owns a
// CONTIGUOUS segment of its row -- four lanes per row, L = ceil(C / 4) elements each -- and reads it as 32-bit words
// from the word its first element lies in; v_alignbit re-aligns the words of an odd start, a shift and a mask widen the
// two halves.  The gradient goes back in place as bf16 halves (v_cvt_pk_bf16_f32 per pair, one 16-bit LDS store per
// element: a word of the tile may belong to two lanes, or two rows) and leaves as flat 16-B stores.
// ---------------------------------------------------------------------------------------
template <int KW, int WPB>
__global__ __launch_bounds__(WPB *WAVE, 4) void mstep_bf16w_kernel(
    const uint16_t *__restrict__ logits, const int64_t *__restrict__ labels, const int64_t *__restrict__ idx,
    const float *__restrict__ weights, float *__restrict__ residuals, int64_t N, int64_t nfull, int C,
    float inv_scale, uint16_t *__restrict__ grad, double *__restrict__ part, int32_t *__restrict__ status,
    int accum, double inv_rows100) {
    constexpr int R = 16, G = 4, NI = 4;                          // 16 rows x <= 128 elements x 2 B <= 4 KiB
    constexpr int WTILE = NI * 1024 + 64;                         // (+ a pad: the word past a segment's end, dummy stores)
    constexpr int NE = 2 * KW;                                    // element slots of a lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int g = lane & (G - 1), sub = lane / G;
    char *wtile = smem + (size_t)wave * WTILE;
    vu4 *tile16 = reinterpret_cast<vu4 *>(wtile);
    const int nchunk = 2 * C;                                     // 16-byte chunks of a tile (16 rows x 2 C bytes)
    const int L = (C + 3) >> 2;
    const int e0 = g * L;
    const int cnt = C - e0 < L ? (C - e0 > 0 ? C - e0 : 0) : L;   // elements of this lane's segment
    const int E0 = sub * C + e0;                                  // first element, counted in the tile
    const unsigned shift = (unsigned)(E0 & 1) * 16u;              // odd start: the segment begins in a word's high half
    const int W0 = E0 >> 1;
    const int64_t *idxp = idx != nullptr ? idx : labels;
    const unsigned sub8 = (unsigned)sub * 8u;
    const int64_t tstride = (int64_t)gridDim.x * WPB;
    float acc = 0.0f, hits = 0.0f;
    bool bad = false;
    for (int64_t t = (int64_t)blockIdx.x * WPB + wave; t < nfull; t += tstride) {
        const int64_t row_base = t * R;
        const char *src = reinterpret_cast<const char *>(logits + row_base * C);
        int64_t y64 = *reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(labels + row_base) + sub8);
        int64_t ix = *reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(idxp + row_base) + sub8);
        vu4 stg[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int c = i * WAVE + lane;
            c = c < nchunk ? c : nchunk - 1;                      // past the tile: re-read its last chunk
            stg[i] = __builtin_nontemporal_load(reinterpret_cast<const vu4 *>(src + (size_t)c * 16));
        }
        bool okrow = true;
        ix = idx != nullptr ? ix : row_base + sub;
        if (y64 < 0 || y64 >= C) { y64 = 0; okrow = false; }
        if (ix < 0 || ix >= N) { ix = 0; okrow = false; }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i * WAVE + lane < nchunk) tile16[i * WAVE + lane] = stg[i];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(y64), "+v"(ix) : : "memory");
        float pi = weights != nullptr ? weights[ix] : 1.0f;         // behind the tile (see mstep_wave_kernel)
        bad = bad || !okrow;
        __builtin_amdgcn_wave_barrier();

        // ---- B: the segment as words, re-aligned, widened
        const int y = (int)y64;
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(wtile);
        const uint16_t *h16 = reinterpret_cast<const uint16_t *>(wtile);
        const float zy = bf16_to_f32(h16[sub * C + y]);
        uint32_t w[KW + 1];
#pragma unroll
        for (int k = 0; k <= KW; ++k) w[k] = w32[W0 + k];
        float v[NE];
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const uint32_t x = __builtin_amdgcn_alignbit(w[k + 1], w[k], shift);      // elements 2k, 2k + 1
            v[2 * k] = __uint_as_float(x << 16);
            v[2 * k + 1] = __uint_as_float(x & 0xFFFF0000u);
        }
        const float NEG_INF = -__builtin_inff();
        float m = NEG_INF;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            v[j] = j < cnt ? v[j] : NEG_INF;
            m = __builtin_fmaxf(m, v[j]);
        }
        m = group_allreduce<G>(m, FMaxF());
        const bool hit0 = zy == m;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            v[j] = mexp(v[j] - m);                                // a slot past the segment: exp(-inf) = 0
            s += v[j];
        }
        s = group_sum<G>(s);
        const float li = __builtin_amdgcn_logf(s) * 0.69314718055994530942f - (zy - m);
        pi = okrow ? pi : 0.0f;
        // top-1: the label is a hit when it is the FIRST column that attains the row maximum; two exact maxima put
        // at least 2.0 into the sum, so the exact check (the words are still in registers) runs only for waves
        // that hold such a row
        bool hit = hit0;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit0 && s >= 2.0f) != 0, 0)) {
            int earlier = 0;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const uint32_t x = __builtin_amdgcn_alignbit(w[k + 1], w[k], shift);
                const float z0 = __uint_as_float(x << 16), z1 = __uint_as_float(x & 0xFFFF0000u);
                earlier += (2 * k < cnt && z0 == m && e0 + 2 * k < y) ? 1 : 0;
                earlier += (2 * k + 1 < cnt && z1 == m && e0 + 2 * k + 1 < y) ? 1 : 0;
            }
            hit = hit0 && group_allreduce<G>(earlier, FAdd()) == 0;
        }
        if (grad != nullptr) {
            const float gs = pi * inv_scale;
            const float inv_s = gs * __builtin_amdgcn_rcpf(s);
            const int jy = y - e0;                                // the label's slot, if it lies in this segment
            uint16_t *o16 = reinterpret_cast<uint16_t *>(wtile);
            const int dummy = NI * 512 + (lane & 15);             // a halfword of the pad, per lane
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                float o0 = v[2 * k] * inv_s, o1 = v[2 * k + 1] * inv_s;
                o0 = 2 * k == jy ? o0 - gs : o0;
                o1 = 2 * k + 1 == jy ? o1 - gs : o1;
                const uint32_t pk = f32x2_to_bf16x2(o0, o1);
                o16[2 * k < cnt ? E0 + 2 * k : dummy] = (uint16_t)(pk & 0xFFFFu);
                o16[2 * k + 1 < cnt ? E0 + 2 * k + 1 : dummy] = (uint16_t)(pk >> 16);
            }
        }
        if (g == 0 && okrow && residuals != nullptr) residuals[ix] = li;
        acc += okrow ? li * pi : 0.0f;
        hits += (hit && okrow) ? 1.0f : 0.0f;
        __builtin_amdgcn_wave_barrier();

        // ---- C: flat store of the gradient tile
        if (grad != nullptr) {
            char *gdst = reinterpret_cast<char *>(grad + row_base * C);
            vu4 st[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) st[i] = tile16[(i * WAVE + lane < nchunk) ? i * WAVE + lane : 0];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (i * WAVE + lane < nchunk)
                    __builtin_nontemporal_store(st[i], reinterpret_cast<vu4 *>(gdst + (size_t)(i * WAVE + lane) * 16));
        }
        __builtin_amdgcn_wave_barrier();
    }
    double a = wave_sum((double)(g == 0 ? acc : 0.0f));
    double h = wave_sum((double)(g == 0 ? hits : 0.0f));
    __shared__ double sh[2 * WPB];
    if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = h; }
    if (bad) atomicOr(status, RLVI_ST_RANGE);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, th = 0.0;
#pragma unroll
        for (int w = 0; w < WPB; ++w) { ta += sh[2 * w]; th += sh[2 * w + 1]; }
        write_partial(part, ta, th, inv_scale, inv_rows100, accum);
    }
}

// Hold of the reads-then-writes form for a launch that reads `bytes` of logits, in ticks of 10 ns: the time
// its reads need at the read-only rate.  Fitted to the best hold of a sweep per shape (tools/sweep_hold.sh:
// 13.6 MB bf16 -> 2.4 us, 16.8 MB -> 2.9, 19.7 MB -> 3.4, 26.2 MB -> 4.3, 33.5 MB -> 5.2): 0.68 us of ramp-up
// + bytes / 7.25 TB/s; within +-0.3 us of the best hold most of the gain stays, 1 us off is worse than
// none.  Below 12 MB no hold was found to help (the launch is over before the phases could separate).
#ifndef RLVI_MSTEP_AUTO_DEFAULT
#define RLVI_MSTEP_AUTO_DEFAULT 0
#endif
static inline int mstep_hold_ticks(double bytes) {
    if (bytes < 12.0e6) return 0;
    return (int)((0.68 + bytes / 7.25e6) * 100.0);
}

__global__ __launch_bounds__(256) void mstep_finalize_kernel(double *__restrict__ part, int nblocks,
                                                             double scale, float *__restrict__ out,
                                                             int clear) {
    reduce_partials(part, nblocks, scale, out, clear != 0, 256);
}

template <typename T, int V, int G, int KMAX>
static int launch_mstep(const T *logits, int64_t ld, const int64_t *labels, const int64_t *idx,
                        const float *weights, float *residuals, int64_t N, int64_t B, int C,
                        int kact, float inv_scale, T *grad, int64_t ldg, float *out, void *ws,
                        hipStream_t st) {
    constexpr int R = WAVE / G;
    constexpr int WPB = 4;                          // waves per workgroup of the wave-tile form
    const int cus = device_info().cus;
    // form: 1 = wave tiles through LDS (dense rows), 0 = register rows everywhere; -1 (default): wave tiles
    // once the launch has more than two waves per CU (512 tiles) -- below that a call is one wave's latency
    // long, and global -> registers -> global is shorter than the trip through LDS (tools/sweep_small.sh:
    // 4096 x 10 3.4 -> 3.1 us, 4096 x 100 4.05 -> 3.7; 16 384 x 100 the other way, 5.2 against 5.7)
    int form = tune_get("RLVI_MSTEP_FORM", -1);
    if (form < 0) form = (B + R - 1) / R > 512 ? 1 : 0;
    char *base = static_cast<char *>(ws);
    // (a call with `out` has records of its own: it never touches what an accumulate sequence has piled up)
    double *part = reinterpret_cast<double *>(base + (out == nullptr ? WS_PART_OFF : WS_PART2_OFF));
    int32_t *status = reinterpret_cast<int32_t *>(base);
    const int accum = out == nullptr ? 1 : 0;       // no `out`: accumulate for rlvi_epoch_end_f32
    const double inv_rows100 = 100.0 / (double)B;
    const bool flat16 = ld == C && (grad == nullptr || ldg == C) &&
                        ((uintptr_t)logits % 16) == 0 && ((uintptr_t)grad % 16) == 0;
    const size_t wtile_bytes = (size_t)R * C * sizeof(T);
    constexpr size_t SKB = (size_t)((KMAX * V * sizeof(T) + 15) / 16);
    const bool dense_wave = flat16 && wtile_bytes % 16 == 0 && wtile_bytes / 16 <= SKB * WAVE;
    int64_t nb;
    int rc;
    const int64_t nfull = B / R;
    if (dense_wave && form >= 1 && nfull > 0) {
        // `wpc` waves per CU stride over the R-row tiles (one tile each at the bench size)
        const int wpc = tune_get("RLVI_MSTEP_WPC", 16);
        nb = (nfull + WPB - 1) / WPB;
        int64_t cap = ((int64_t)wpc * cus + WPB - 1) / WPB;
        if (cap > MSTEP_MAX_BLOCKS) cap = MSTEP_MAX_BLOCKS;
        if (nb > cap) nb = cap;
        // (RLVI_MSTEP_LDS_PAD: extra LDS per workgroup = fewer resident workgroups per CU; with the grid
        //  uncapped the dispatcher then hands the remaining tiles to whichever CU frees up first)
        const size_t lds = (size_t)WPB * SKB * 1024 + (size_t)tune_get("RLVI_MSTEP_LDS_PAD", 0);
        // reads-then-writes hold (see the kernel): only when no wave has a second tile and a gradient is
        // written.  RLVI_MSTEP_HOLD = ticks of 10 ns; 0 (the default): off; -1: from the bytes the launch
        // reads AT THE HBM READ RATE -- for logits that stream from HBM (a block that is not resident in the
        // Infinity Cache: bench.py's rotation of twelve blocks, a block larger than the cache).  Logits that
        // the model's last layer has just written are served by the cache, their read phase is over sooner
        // than the hold assumes, and the hold then COSTS time (single buffer pair: 9.4 -> 10.3 us): the
        // caller knows which case it is in, the kernel does not (a per-CU barrier between reads and writes
        // -- 16-wave workgroups -- was built to let the data decide: no gain cold, 9.4 -> 10.1 us warm).
        // (the caller's hint lives with the caller's workspace -- rlvi_workspace_set_option(ws, "logits_from_hbm", 1)
        //  -- not with the process: two training loops, or two streams, do not see each other's; the knob of the
        //  same meaning is the lab override)
        int hold_ticks = tune_get("RLVI_MSTEP_HOLD", ws_option(ws, WSOPT_LOGITS_FROM_HBM, 0) ? -1 : 0);
        int gen_ticks = tune_get("RLVI_MSTEP_GEN", -1);      // ticks between the holds of successive tile generations
        const int64_t waves = nb * WPB;
        const double gen_bytes = (double)(nfull < waves ? nfull : waves) * (double)wtile_bytes;
        if (nfull > waves && hold_ticks < 0 && gen_ticks < 0) {
            // several tiles per wave under the caller's HBM hint: one hold per GENERATION of tiles, the
            // generations one read + one write of their bytes at the mixed rate apart (2 x 26.2 MB: 9.0 us).
            // Only where it was measured to pay (tools/sweep_gen.sh): fp32, two to four FULL generations
            // (65 536 x 100 per generation: 21.8 -> 20.0 us at two, 30.6 -> 29.3 at three, 39.8 -> 39.0 at four;
            // five gain or lose a per cent with the spacing, a half-filled last generation or eight lose)
            const int64_t gens = nfull / waves;
            gen_ticks = (sizeof(T) == 4 && nfull % waves == 0 && gens >= 2 && gens <= 4 && gen_bytes >= 12.0e6)
                            ? (int)(2.0 * gen_bytes / 5.76e6 * 100.0) : 0;
        }
        if (gen_ticks < 0) gen_ticks = 0;
        if (grad == nullptr || (nfull > waves && gen_ticks <= 0 && hold_ticks < 0)) hold_ticks = 0;
        if (hold_ticks < 0) hold_ticks = mstep_hold_ticks(gen_bytes);
        if (hold_ticks == 0) gen_ticks = 0;
        // the self-timed hold (see the kernel): a chip-filling launch with one tile per wave that writes a gradient
        // and reads at least 12 MB (below that no hold was found to help), unless the caller or the lab said
        // something else.  RLVI_MSTEP_AUTO=0: the 16-wave barrier form of round 3 instead.
        unsigned long long *hold_slot = nullptr;
        unsigned long long hold_key = 0ull;
        int hold_cap = 0;
        const int hold_pct = tune_get("RLVI_MSTEP_AUTO_PCT", 100);
        if (tune_get("RLVI_MSTEP_AUTO", RLVI_MSTEP_AUTO_DEFAULT) && grad != nullptr && hold_ticks == 0 &&
            nfull <= waves && gen_bytes >= 12.0e6) {
            hold_key = ((unsigned long long)nfull << 32) ^ ((unsigned long long)C * sizeof(T) << 8) ^ (unsigned)WPB;
            WsHeader *hdr = reinterpret_cast<WsHeader *>(base);
            hold_slot = &hdr->mstep_hold[(unsigned)((nfull * 31 + C * (int)sizeof(T)) & 3)][0];
            hold_cap = mstep_hold_ticks(gen_bytes) * 3 / 2;      // (a stamp of a descheduled wave must not stall a launch)
        }
        bool cuwide_done = false;
        if constexpr (G == 4 && V * sizeof(T) == 16) {
            constexpr int WPB16 = 16;
            int64_t nb16 = (nfull + WPB16 - 1) / WPB16;
            // (a caller that has hinted HBM-resident logits gets the four-wave workgroups with the timed hold:
            //  10.65 against 10.95 us -- the timed hold separates the phases chip-wide, the barrier per CU)
            if (tune_get("RLVI_MSTEP_CUWIDE", 1) && grad != nullptr && hold_ticks == 0 && hold_slot == nullptr &&
                nb16 <= cus &&
                nb16 * 5 >= (int64_t)cus * 4 && nb16 <= MSTEP_MAX_BLOCKS) {
                const size_t lds16 = (size_t)WPB16 * SKB * 1024;
                auto go = [&](auto kern) {
                    static int attr_dev = -1;      // (> 64 KiB of dynamic LDS: asked for once per kernel and device)
                    int cur_dev = 0;
                    if (hipGetDevice(&cur_dev) != hipSuccess) return (int)hipErrorInvalidDevice;
                    if (attr_dev != cur_dev) {
                        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
                        if (e != hipSuccess) return (int)e;
                        attr_dev = cur_dev;
                    }
                    return launch(kern, dim3((unsigned)nb16), dim3(WPB16 * WAVE), lds16, st, logits, labels, idx,
                                  weights, residuals, N, nfull, C, inv_scale, grad, part, status, accum, inv_rows100,
                                  hold_ticks, 0, (unsigned long long *)nullptr, 0ull, 0, 100);
                };
                rc = kact == KMAX ? go(mstep_wave_kernel<T, V, G, KMAX, WPB16, true>)
                                  : go(mstep_wave_kernel<T, V, G, KMAX, WPB16, false>);
                nb = nb16;
                cuwide_done = true;
                ws_note_mstep(ws, 3);
            }
        }
        if (!cuwide_done) ws_note_mstep(ws, 2 + (hold_ticks != 0 ? 16 : 0) + (hold_slot != nullptr ? 32 : 0));
        if (cuwide_done) {
        } else if (kact == KMAX)
            rc = launch(mstep_wave_kernel<T, V, G, KMAX, WPB, true>, dim3((unsigned)nb), dim3(WPB * WAVE),
                        lds, st, logits, labels, idx, weights, residuals, N, nfull, C, inv_scale, grad,
                        part, status, accum, inv_rows100, hold_ticks, gen_ticks, hold_slot, hold_key, hold_cap, hold_pct);
        else
            rc = launch(mstep_wave_kernel<T, V, G, KMAX, WPB, false>, dim3((unsigned)nb), dim3(WPB * WAVE),
                        lds, st, logits, labels, idx, weights, residuals, N, nfull, C, inv_scale, grad,
                        part, status, accum, inv_rows100, hold_ticks, gen_ticks, hold_slot, hold_key, hold_cap, hold_pct);
        const int64_t done = nfull * R;
        if (rc == 0 && done < B) {
            // the B mod R trailing rows: one workgroup of the register-row kernel, adding to record 0
            rc = launch(mstep_kernel<T, V, G, KMAX>, dim3(1), dim3(MSTEP_THREADS), 0, st,
                        logits + done * ld, ld, labels + done, idx != nullptr ? idx + done : idx, weights,
                        residuals, N, B - done, C, kact, inv_scale, grad != nullptr ? grad + done * ldg : grad,
                        ldg, part, status, 1, inv_rows100, done);
        }
    } else {
        const int max_blocks = tune_get("RLVI_MSTEP_BLOCKS", MSTEP_MAX_BLOCKS);
        const int64_t rows_per_block = (int64_t)MSTEP_WAVES * R;
        nb = (B + rows_per_block - 1) / rows_per_block;
        if (nb > max_blocks) nb = max_blocks;
        if (nb > MSTEP_MAX_BLOCKS) nb = MSTEP_MAX_BLOCKS;
        if (nb < 1) nb = 1;
        ws_note_mstep(ws, 1);
        rc = launch(mstep_kernel<T, V, G, KMAX>, dim3((unsigned)nb), dim3(MSTEP_THREADS), 0, st,
                    logits, ld, labels, idx, weights, residuals, N, B, C, kact, inv_scale, grad, ldg,
                    part, status, accum, inv_rows100, (int64_t)0);
    }
    if (rc != 0 || out == nullptr) return rc;
    // the finalize pass clears what it read: records are all-zero outside an accumulate sequence
    return launch(mstep_finalize_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, 1.0, out, 1);
}

// Picks the lane group: the smallest G whose lanes need at most 8 vectors each (so the short
// DPP reductions are amortised over up to 8*V elements per lane); rows of <= 8 vectors are
// handled by a single lane.
template <typename T, int V>
static int dispatch_gk(const T *logits, int64_t ld, const int64_t *labels, const int64_t *idx,
                       const float *weights, float *residuals, int64_t N, int64_t B, int C,
                       float inv_scale, T *grad, int64_t ldg, float *out, void *ws,
                       hipStream_t st) {
    const int nv = (C + V - 1) / V;
    const int force_g = tune_get("RLVI_MSTEP_G", 0);
#define RLVI_CASE(G_, K_)                                                                        \
    return launch_mstep<T, V, G_, K_>(logits, ld, labels, idx, weights, residuals, N, B, C,      \
                                      (nv + G_ - 1) / G_, inv_scale, grad, ldg, out, ws, st)
    // (bf16 in 16-byte vectors: at most FOUR vectors = 32 elements per lane, as fp32's eight -- with 64 elements a lane's
    //  serial work and its registers cost more than the shorter reductions save: tools/lab/sweep_g_bf16.sh, 16 384 x 256
    //  11.2 -> 6.5 us, x 512 15.8 -> 8.6, x 1000 24.3 -> 15.8, x 2000 39.5 -> 26.7, 4096 x 1000 10.8 -> 6.6)
    //  Rows of up to 20 vectors keep the 16-row tile with four lanes per row: five vectors per lane there beat three on
    //  eight lanes, 65 536 x 136 10.6 against 13.3 us, x 160 11.1 against 13.7; x 200 -- seven -- 15.7 against 14.4.)
    constexpr bool WIDE_BF16 = sizeof(T) == 2 && V == 8;
    const int kpref = (WIDE_BF16 && nv > 20) ? 4 : 8;
    int gsel = 64;
    for (int gg = 1; gg <= 64; gg <<= 1)
        if ((nv + gg - 1) / gg <= kpref) { gsel = gg; break; }
    // 64-row tiles (G = 4) measured best whenever a row has at least 8 vectors (fp32 C = 100:
    // 10.8 us against 11.8 for G = 8; bf16 C = 104: 8.7 us against 12.3 for G = 2)
    if (gsel < 4 && nv >= 8) gsel = 4;
    if (force_g && (nv + force_g - 1) / force_g <= 8) gsel = force_g;
    // rows of up to 128 elements whose length is not a multiple of the 16-byte vector (C = 101:
    // V = 1) keep the 64-row tile and four lanes per row, each lane holding up to 32 / V short
    // vectors: 11.6 us instead of 20.0 at 65 536 x 101 fp32 (the 16-lane form spends its time in
    // per-element address arithmetic and DPP steps)
    // -- only for launches of at least two such tiles per SIMD (fp32 C = 101: 8.6 against 9.4 us at 32 768
    // rows, 7.2 against 5.8 at 16 384, 5.7 against 3.1 at 1024), and never for single 2-byte elements (bf16
    // C = 101: one ds_read_u16 per element, 27.5 against 14.3 us at 65 536 rows, 9.2 against 3.1 at 1024)
    if constexpr (V * sizeof(T) < 16 && V * sizeof(T) > 2) {
        if (gsel > 4 && C <= 128 && !force_g && B >= 16 * 2048) {
            const int k4 = (nv + 3) / 4;
            if (k4 <= 16) RLVI_CASE(4, 16);
            if constexpr (V == 1) RLVI_CASE(4, 32);
        }
        // the same for rows of 129 ... 512 such elements (Places365's 365 classes): sixteen lanes per row, four rows per
        // wave tile, instead of one row per wave in single 4-byte loads (lab knob RLVI_MSTEP_ODD16=0: the old route)
        // -- 65 536 x 365 fp32 46.0 -> 39.5 us, 16 384 x 365 15.5 -> 11.8, 65 536 x 201 27.9 -> 22.1, 65 536 x 366 bf16
        // 30.9 -> 24.4; not below 8192 rows (4096 x 365: 5.9 against 7.2) and not beyond 24 single elements per lane (x 511:
        // 54.6 against 58.0, the 32-slot form spills)
        if (gsel > 16 && C > 128 && C <= 512 && !force_g && B >= tune_get("RLVI_MSTEP_ODD16_ROWS", 8192) &&
            tune_get("RLVI_MSTEP_ODD16", 1)) {
            const int k16 = (nv + 15) / 16;
            if constexpr (!(sizeof(T) == 2 && V == 4)) {         // (bf16 in 8-byte vectors never has gsel > 16 here)
                if (k16 <= 16) RLVI_CASE(16, 16);
            }
            if constexpr (V == 1) {
                if (k16 <= 24) RLVI_CASE(16, 24);
            }
        }
    }
    // a launch of less than one wave per SIMD is as long as ONE wave's instruction stream (a lone wave issues
    // an instruction every ~6 clocks): twice the lanes per row halve it (tools/sweep_small.sh: 4096 x 10
    // 3.65 -> 3.3 us, 4096 x 100 4.4 -> 4.05, 1024 x 101 3.35 -> 3.1; from 1024 waves on the wider tile wins)
    if (!force_g && kpref == 8 && gsel < 64 && (nv + gsel - 1) / gsel >= 2 && (B * gsel + 63) / 64 < 1024) gsel *= 2;
    // bf16 rows of five to eight 16-byte vectors (C = 40 ... 64) from 16 384 rows on: two lanes per row (32 rows per wave
    // tile) -- 65 536 x 64 8.6 -> 7.1 us, x 48 10.6 -> 7.0, x 40 10.5 -> 6.8, 262 144 x 64 22.5 -> 18.2; nine vectors (x 72) are
    // better off with four lanes, four (x 32) with one.  (fp32 C = 20 ... 32 would gain 8-12 % too -- 65 536 x 32 6.75 -> 6.2 us --
    // but the one-launch in-batch E+M promises the bits of this kernel's four-lane row sums at those shapes.)
    if (!force_g && sizeof(T) == 2 && V == 8 && nv >= 5 && nv <= 8 && B >= 16384) gsel = 2;
    const int k = (nv + gsel - 1) / gsel;
    if (k > 8) {
        // more than 512 vectors per row: a wave (or, for few rows, a workgroup) per row, three passes (mstep_longrow_kernel)
        char *base = static_cast<char *>(ws);
        double *part = reinterpret_cast<double *>(base + (out == nullptr ? WS_PART_OFF : WS_PART2_OFF));
        // fewer rows than four per CU: the whole workgroup on one row
        const bool wide = B <= 4 * (int64_t)device_info().cus;
        int64_t nb = wide ? B : (B + MSTEP_WAVES - 1) / MSTEP_WAVES;
        if (nb > MSTEP_MAX_BLOCKS) nb = MSTEP_MAX_BLOCKS;
        ws_note_mstep(ws, 5);
        const int rc = wide ? launch(mstep_longrow_kernel<T, V, MSTEP_WAVES>, dim3((unsigned)nb), dim3(MSTEP_THREADS), 0, st,
                                     logits, ld, labels, idx, weights, residuals, N, B, C, inv_scale, grad, ldg, part,
                                     reinterpret_cast<int32_t *>(base), out == nullptr ? 1 : 0, 100.0 / (double)B)
                            : launch(mstep_longrow_kernel<T, V, 1>, dim3((unsigned)nb), dim3(MSTEP_THREADS), 0, st,
                                     logits, ld, labels, idx, weights, residuals, N, B, C, inv_scale, grad, ldg, part,
                                     reinterpret_cast<int32_t *>(base), out == nullptr ? 1 : 0, 100.0 / (double)B);
        if (rc != 0 || out == nullptr) return rc;
        return launch(mstep_finalize_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, 1.0, out, 1);
    }
    switch (gsel) {
        case 1:
            if (k <= 1) RLVI_CASE(1, 1);
            if (k <= 2) RLVI_CASE(1, 2);
            if (k <= 4) RLVI_CASE(1, 4);
            RLVI_CASE(1, 8);
        case 2: if (k <= 4) RLVI_CASE(2, 4); RLVI_CASE(2, 8);
        case 4:
            if (k <= 4) RLVI_CASE(4, 4);
            if (k == 5) RLVI_CASE(4, 5);
            if (k == 6) RLVI_CASE(4, 6);
            if (k == 7) RLVI_CASE(4, 7);
            RLVI_CASE(4, 8);
        case 8: if (k <= 2) RLVI_CASE(8, 2); if (k <= 4) RLVI_CASE(8, 4); RLVI_CASE(8, 8);
        case 16: if (k <= 4) RLVI_CASE(16, 4); RLVI_CASE(16, 8);
        case 32: if (k <= 4) RLVI_CASE(32, 4); RLVI_CASE(32, 8);
        default: if (k <= 4) RLVI_CASE(64, 4); RLVI_CASE(64, 8);
    }
#undef RLVI_CASE
}

template <typename T>
static int mstep_entry(const T *logits, int64_t ld, const int64_t *labels, const int64_t *idx,
                       const float *weights, float *residuals, int64_t N, int64_t B, int64_t C,
                       float inv_scale, T *grad, int64_t ldg, float *out, void *ws, void *stream) {
    if (!logits || !labels || !ws) return RLVI_E_NULL;
    if (!weights && idx) return RLVI_E_NULL;       // an index without the vector it indexes
    if (!weights) N = B;                            // evaluation form: identity rows, pi = 1
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
    if constexpr (sizeof(T) == 2) {
        // bf16 rows of an odd number of elements (single 2-byte elements in the general forms), at least 32 768 of
        // them, dense, 16-byte aligned: the
        // word-wise wave tile (mstep_bf16w_kernel); the B mod 16 trailing rows go to the register-row kernel
        if (((C & 1) != 0 || tune_get("RLVI_MSTEP_BF16W", 1) == 2) && C >= 9 && C <= 127 && B >= 16 * 2048 && ld == C && (grad == nullptr || ldg == C) &&
            ((uintptr_t)logits % 16) == 0 && ((uintptr_t)grad % 16) == 0 && tune_get("RLVI_MSTEP_BF16W", 1) &&
            tune_get("RLVI_MSTEP_FORM", -1) != 0 && tune_get("RLVI_MSTEP_G", 0) == 0) {
            constexpr int WPB = 4;
            char *base = static_cast<char *>(ws);
            double *part = reinterpret_cast<double *>(base + (out == nullptr ? WS_PART_OFF : WS_PART2_OFF));
            int32_t *status = reinterpret_cast<int32_t *>(base);
            const int accum = out == nullptr ? 1 : 0;
            const double inv_rows100 = 100.0 / (double)B;
            const int64_t nfull = B / 16;
            int64_t nb = (nfull + WPB - 1) / WPB;
            int64_t cap = ((int64_t)tune_get("RLVI_MSTEP_WPC", 16) * device_info().cus + WPB - 1) / WPB;
            if (cap > MSTEP_MAX_BLOCKS) cap = MSTEP_MAX_BLOCKS;
            if (nb > cap) nb = cap;
            const size_t lds = (size_t)WPB * (4 * 1024 + 64);
            const int L = (Ci + 3) / 4;
            const int kw = (L + 1) / 2;
            int rc;
            ws_note_mstep(ws, 4);
#define RLVI_BW(KW_)                                                                                       \
    rc = launch(mstep_bf16w_kernel<KW_, WPB>, dim3((unsigned)nb), dim3(WPB * WAVE), lds, st, logits, labels, idx, \
                weights, residuals, N, nfull, Ci, inv_scale, grad, part, status, accum, inv_rows100)
            if (kw <= 8) RLVI_BW(8);
            else if (kw <= 13) RLVI_BW(13);
            else RLVI_BW(16);
#undef RLVI_BW
            const int64_t done = nfull * 16;
            if (rc == 0 && done < B)
                rc = launch(mstep_kernel<T, 1, 16, 8>, dim3(1), dim3(MSTEP_THREADS), 0, st, logits + done * ld, ld,
                            labels + done, idx != nullptr ? idx + done : idx, weights, residuals, N, B - done, Ci,
                            (Ci + 15) / 16, inv_scale, grad != nullptr ? grad + done * ldg : grad, ldg, part, status, 1,
                            inv_rows100, done);
            if (rc != 0 || out == nullptr) return rc;
            return launch(mstep_finalize_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, 1.0, out, 1);
        }
    }
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

// Reduce (and clear) the partial records that accumulate-mode M-step calls left in the workspace.
extern "C" int rlvi_mstep_reduce_f32(float *out, double scale, void *ws, void *stream) {
    if (!out || !ws) return RLVI_E_NULL;
    double *part = reinterpret_cast<double *>(static_cast<char *>(ws) + rlvi::WS_PART_OFF);
    return rlvi::launch(rlvi::mstep_finalize_kernel, dim3(1), dim3(256), 0,
                        static_cast<hipStream_t>(stream), part, (int)rlvi::MSTEP_MAX_BLOCKS, scale, out, 1);
}
