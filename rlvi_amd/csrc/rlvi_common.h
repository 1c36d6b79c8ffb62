// Shared device helpers for librlvi_gfx950.so (gfx950 only: 64-lane waves, 256 CUs, 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rlvi_hip.h"

namespace rlvi {

constexpr int WAVE = 64;
constexpr int NUM_CU = 256;
constexpr int MAX_COOP_WG = 256;        // one workgroup per CU for the cooperative kernels

// ---------------------------------------------------------------------------------------
// Workspace layout (bytes).  Control words first, each group on its own 256-B line.
// ---------------------------------------------------------------------------------------
struct WsHeader {
    int32_t status;          // sticky RLVI_ST_* flags
    uint32_t pad0[63];
    uint32_t mstep_ticket;   // last-block-done counter of the M-step kernel (self-resetting)
    uint32_t pad1[63];
    uint32_t epoch_base;     // tag base of the exchange slots (advanced by every coop kernel)
    uint32_t pad2[63];
};
constexpr size_t WS_HDR_BYTES = sizeof(WsHeader);                       // 768
constexpr size_t WS_XCHG_OFF = 1024;
constexpr int XCHG_GRANULES = 4;                                        // per workgroup
constexpr size_t WS_XCHG_BYTES = 2ull * MAX_COOP_WG * XCHG_GRANULES * 8; // 2 parities, 16 KiB
constexpr size_t WS_PART_OFF = WS_XCHG_OFF + WS_XCHG_BYTES;
constexpr int MSTEP_MAX_BLOCKS = 2048;
constexpr size_t WS_PART_BYTES = (size_t)MSTEP_MAX_BLOCKS * 2 * 8;      // {sum pi*l, hits} per block
constexpr size_t WS_SCRATCH_OFF = WS_PART_OFF + WS_PART_BYTES;

__host__ __device__ inline size_t ws_bytes_for(int64_t max_n, int64_t max_b) {
    // scratch: two fp32 vectors of max(max_n, max_b) (fused E+M keeps l and e there)
    int64_t m = max_n > max_b ? max_n : max_b;
    if (m < 0) m = 0;
    size_t s = WS_SCRATCH_OFF + (size_t)m * 8 + 256;
    return (s + 255) & ~(size_t)255;
}

// ---------------------------------------------------------------------------------------
// Wave (64-lane) butterfly reductions.  a+b is commutative in IEEE arithmetic, so after
// the butterfly every lane holds bit-identical totals, in a fixed association order.
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        T u = __shfl_xor(v, o, WAVE);
        v = u > v ? u : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        T u = __shfl_xor(v, o, WAVE);
        v = u < v ? u : v;
    }
    return v;
}

// Reductions inside a lane group of G consecutive lanes (G a power of two <= 64).
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <int G>
__device__ __forceinline__ int group_min_i(int v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
        int u = __shfl_xor(v, o, WAVE);
        v = u < v ? u : v;
    }
    return v;
}

// Order-preserving key of an fp32 value (ascending value <=> ascending unsigned key).
__device__ __forceinline__ uint32_t f32_key(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __uint_as_float((uint32_t)h << 16);
}
// round-to-nearest-even, NaN kept NaN (plain cast semantics)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace rlvi
