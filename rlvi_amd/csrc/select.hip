// Small-loss selection: the k smallest of n per-sample losses as a 0/1 weight vector.
//
// Replaces the host-side  ind_sorted = np.argsort(loss.cpu()); ind_update = ind_sorted[:k]  of the
// small-loss baselines (train_usdnl.py:18-24, train_coteaching.py:18-30): the selected rows then
// get weight 1 in the streaming M-step kernel (mstep.hip) instead of being gathered into a new
// batch, so the loss/gradient pass is the same kernel the RLVI path uses.
//
// One workgroup, no sort: a 4-pass radix select on the order-preserving key (8 bits per pass, LDS
// histogram, the bin holding the k-th element fixes the next byte) finds the key T of the k-th
// smallest loss; rows with key < T are selected, and of the rows with key == T the first
// (k - #{key < T}) in index order -- the order a stable argsort gives (numpy's default sort is not
// stable; which of several EQUAL losses it keeps is an implementation detail of the reference).
// NaN losses order last, as in numpy.  (-0.0 orders before +0.0; losses are >= 0.)
#include "rlvi_common.h"

namespace rlvi {

constexpr int SEL_BLOCK = 1024;
constexpr int SEL_NW = SEL_BLOCK / WAVE;

__global__ __launch_bounds__(SEL_BLOCK) void select_smallest_kernel(const float *__restrict__ loss,
                                                                    int64_t n, int64_t k,
                                                                    float *__restrict__ mask_w) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_need, sh_ties;
    __shared__ unsigned wcount[SEL_NW];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    if (k <= 0 || k >= n) {                      // nothing / everything selected
        const float v = k <= 0 ? 0.0f : 1.0f;
        for (int64_t i = tid; i < n; i += SEL_BLOCK) mask_w[i] = v;
        return;
    }
    if (tid == 0) { sh_prefix = 0u; sh_need = (unsigned)k; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sh_prefix;
        for (int64_t i = tid; i < n; i += SEL_BLOCK) {
            const unsigned key = f32_key(loss[i]);
            // candidates: keys that agree with the prefix in the bytes fixed so far
            if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned need = sh_need, cum = 0u;
            int b = 0;
            for (; b < 255; ++b) {               // the bin holding the need-th candidate
                const unsigned h = hist[b];
                if (cum + h >= need) break;
                cum += h;
            }
            sh_need = need - cum;
            sh_prefix = prefix | ((unsigned)b << shift);
            sh_ties = hist[b];                   // after the last pass: rows with key == T
        }
        __syncthreads();
    }
    const unsigned T = sh_prefix, need = sh_need, ties = sh_ties;
    if (ties == need) {                          // all rows equal to T are taken: no ranking
        for (int64_t i = tid; i < n; i += SEL_BLOCK) mask_w[i] = f32_key(loss[i]) <= T ? 1.0f : 0.0f;
        return;
    }
    // ties beyond the quota: take the first `need` of them in index order
    unsigned running = 0u;
    for (int64_t base = 0; base < n; base += SEL_BLOCK) {
        const int64_t i = base + tid;
        const unsigned key = i < n ? f32_key(loss[i]) : 0xFFFFFFFFu;
        const bool tie = i < n && key == T;
        const unsigned long long bal = __ballot(tie);
        const unsigned before = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned off = running, total = 0u;
#pragma unroll
        for (int w = 0; w < SEL_NW; ++w) {
            const unsigned c = wcount[w];
            if (w < wave) off += c;
            total += c;
        }
        if (i < n) mask_w[i] = (key < T || (tie && off + before < need)) ? 1.0f : 0.0f;
        running += total;
        __syncthreads();
    }
}

}  // namespace rlvi

using namespace rlvi;

extern "C" int rlvi_select_smallest_f32(const float *loss, int64_t n, int64_t k, float *mask_w,
                                        void *stream) {
    if (!loss || !mask_w) return RLVI_E_NULL;
    if (n < 0 || k < 0) return RLVI_E_SHAPE;
    if (n >= (int64_t)1 << 31) return RLVI_E_LIMIT;
    if (((uintptr_t)loss & 3) || ((uintptr_t)mask_w & 3)) return RLVI_E_ALIGN;
    if (n == 0) return 0;
    return launch(select_smallest_kernel, dim3(1), dim3(SEL_BLOCK), 0, static_cast<hipStream_t>(stream),
                  loss, n, k, mask_w);
}
