// precision@k: how many rows of a batch have their label among the k largest logits.
//
// Replaces accuracy(logit, target, topk) of deep-learning/utils.py:65-79 (softmax :67, torch.topk :70, eq :72,
// the per-k counts :76-77).  train_rlvi discards everything but precision@1 (train_rlvi.py:85, which the
// streaming M-step kernel produces on the side); this is the stand-alone form for callers that want @5.
// No softmax, no sort: the rank of the label's logit inside its row is a count,
//     rank_i = #{c : z_ic > z_iy} + #{c < y : z_ic == z_iy},     hit@k = rank_i < k,
// one pass over the logits, 64 / G rows per wave, the counts per k added with integer atomics (order-free: the
// same bits every run).  Equal values rank in column order; which of several equal values torch.topk lists
// first is an implementation detail of the reference, as are logits that differ but whose fp32 softmax
// values coincide (unpinned; tests/golden/g10_topk.npz holds tie-free rows).
#include "rlvi_common.h"

namespace rlvi {

constexpr int TOPK_THREADS = 1024;         // sixteen waves per workgroup: few workgroups, few atomics on the counters
constexpr int TOPK_WAVES = TOPK_THREADS / WAVE;
constexpr int TOPK_MAXK = 8;              // k values per call

struct TopkList { int k[TOPK_MAXK]; };

__device__ __forceinline__ float topk_widen(float v) { return v; }
__device__ __forceinline__ float topk_widen(uint16_t v) { return bf16_to_f32(v); }

// G = lanes per row (a power of two, chosen by the launcher so that a lane holds at most eight elements of rows up
// to 512 columns): 64 / G rows per wave at once, their elements asked for BEFORE the label's logit is known (the
// label -> logit chain is two dependent round trips), longer rows streamed behind it.
template <typename T>
__global__ __launch_bounds__(TOPK_THREADS, 1) void topk_hits_kernel(const T *__restrict__ logits, int64_t ld,
                                                                 const int64_t *__restrict__ labels, int64_t B, int C,
                                                                 int G, TopkList ks, int nk, int32_t *__restrict__ hits) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int R = WAVE / G;
    const int g = lane & (G - 1), sub = lane / G;
    int mine[TOPK_MAXK];
#pragma unroll
    for (int j = 0; j < TOPK_MAXK; ++j) mine[j] = 0;
    const bool in_regs = C <= 8 * G;
    const int64_t stride = (int64_t)gridDim.x * TOPK_WAVES * R;
    for (int64_t row0 = ((int64_t)blockIdx.x * TOPK_WAVES + wave) * R; row0 < B; row0 += stride) {
        const int64_t row = row0 + sub;
        const bool valid = row < B;
        const T *z = logits + (valid ? row : B - 1) * ld;
        const int64_t y64 = labels[valid ? row : B - 1];
        float v[8];
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = g + j * G;
                v[j] = topk_widen(z[c < C ? c : g < C ? g : 0]);
            }
        }
        const bool ok = valid && y64 >= 0 && y64 < C;   // a label outside [0, C) matches no prediction (utils.py:72)
        const int y = ok ? (int)y64 : 0;
        int ahead = 0;
        float zy;
        if (in_regs) {
            // the label's logit is already in the registers of one lane of the row's group: no second, dependent load
            const int jy = y / G, gy = y & (G - 1);
            float cand = v[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) cand = j == jy ? v[j] : cand;
            zy = __shfl(cand, sub * G + gy, WAVE);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = g + j * G;
                ahead += (c < C && (v[j] > zy || (v[j] == zy && c < y))) ? 1 : 0;
            }
        } else {
            zy = topk_widen(z[y]);
            for (int c = g; c < C; c += G) {
                const float x = topk_widen(z[c]);
                ahead += (x > zy || (x == zy && c < y)) ? 1 : 0;
            }
        }
        for (int m = 1; m < G; m <<= 1) ahead += __shfl_xor(ahead, m, WAVE);
#pragma unroll
        for (int j = 0; j < TOPK_MAXK; ++j) mine[j] += (j < nk && g == 0 && ok && ahead < ks.k[j]) ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < TOPK_MAXK; ++j) mine[j] = group_allreduce<WAVE>(mine[j], FAdd());
    __shared__ int sh[TOPK_WAVES][TOPK_MAXK];
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < TOPK_MAXK; ++j) sh[wave][j] = mine[j];
    }
    __syncthreads();
    // one 64-bit add per PAIR of counters where the array allows it (counts stay below 2^31: no carry between halves)
    const bool pairs = ((uintptr_t)hits & 7) == 0;
    if (threadIdx.x < nk) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < TOPK_WAVES; ++w) t += sh[w][threadIdx.x];
        const int j = threadIdx.x;
        const int tn = __shfl_down(t, 1, WAVE);                       // (nk <= 8: all in wave 0)
        if (pairs && (j & 1) == 0 && j + 1 < nk) {
            const unsigned long long both = (unsigned long long)(unsigned)t | ((unsigned long long)(unsigned)tn << 32);
            if (both) atomicAdd(reinterpret_cast<unsigned long long *>(hits + j), both);
        } else if (!(pairs && (j & 1) == 1) && t) {
            atomicAdd(hits + j, t);
        }
    }
}

template <typename T>
static int topk_entry(const T *logits, int64_t ld, const int64_t *labels, int64_t B, int64_t C, const int32_t *ks,
                      int nk, int32_t *hits, void *stream) {
    if (!logits || !labels || !ks || !hits) return RLVI_E_NULL;
    if (B <= 0 || C <= 0 || ld < C || nk < 1 || nk > TOPK_MAXK) return RLVI_E_SHAPE;
    if (C > (1 << 30)) return RLVI_E_LIMIT;
    if (((uintptr_t)labels & 7) || ((uintptr_t)hits & 3) || ((uintptr_t)logits % sizeof(T))) return RLVI_E_ALIGN;
    TopkList kl;
    for (int j = 0; j < TOPK_MAXK; ++j) kl.k[j] = j < nk ? ks[j] : 0;
    for (int j = 0; j < nk; ++j)
        if (kl.k[j] < 1 || kl.k[j] > C) return RLVI_E_SHAPE;      // torch.topk raises beyond the row length
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(hits, 0, (size_t)nk * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    int G = 1;
    while (G < WAVE && C > 8 * (int64_t)G) G <<= 1;
    const int64_t rows_per_block = (int64_t)TOPK_WAVES * (WAVE / G);
    int64_t nb = (B + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)device_info().cus;
    if (nb > cap) nb = cap;
    return launch(topk_hits_kernel<T>, dim3((unsigned)nb), dim3(TOPK_THREADS), 0, st, logits, ld, labels, B, (int)C, G,
                  kl, nk, hits);
}

}  // namespace rlvi

extern "C" int rlvi_topk_hits_f32(const float *logits, int64_t ld, const int64_t *labels, int64_t B, int64_t C,
                                  const int32_t *ks, int nk, int32_t *hits, void *stream) {
    return rlvi::topk_entry<float>(logits, ld, labels, B, C, ks, nk, hits, stream);
}

extern "C" int rlvi_topk_hits_bf16(const uint16_t *logits, int64_t ld, const int64_t *labels, int64_t B, int64_t C,
                                   const int32_t *ks, int nk, int32_t *hits, void *stream) {
    return rlvi::topk_entry<uint16_t>(logits, ld, labels, B, C, ks, nk, hits, stream);
}
