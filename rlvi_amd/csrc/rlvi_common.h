// Shared device helpers for librlvi_gfx950.so (gfx950 only: 64-lane waves, 256 CUs, 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#include "rlvi_hip.h"

// In-kernel wall-clock stamps and value dumps of the cooperating kernels (RLVI_TJ_DEBUG / RLVI_THR_DEBUG
// at run time) exist only in a -DRLVI_STAMPS=1 build (tools/build_variants.py stamps; load it with
// RLVI_LIB_PATH): in the product build they are compiled out -- the stamp pointer, its counter and the
// guards around every dump cost the trajectory kernels some twenty registers, i.e. a wave per SIMD.
#ifndef RLVI_STAMPS
#define RLVI_STAMPS 0
#endif

namespace rlvi {

constexpr int WAVE = 64;
constexpr int NUM_CU_DEFAULT = 256;     // MI355X; launchers ask the runtime (device_info())
constexpr int MAX_COOP_WG = 256;        // exchange slots in the workspace: at most this many exchanging workgroups

// ---- host side (devinfo.hip) -------------------------------------------------------------
struct DeviceInfo {
    int cus;          // compute units of the current device
    int lds_per_cu;   // bytes
};
const DeviceInfo &device_info();
// Integer knob: rlvi_tune_set() value, else the environment variable of that name, else dflt.
int tune_get(const char *name, int dflt);
// Per-workspace launch option (rlvi_workspace_set_option), else dflt.  Host side only: the launchers ask it.
enum { WSOPT_LOGITS_FROM_HBM = 0, WSOPT_COLD_START = 1, WSOPT_COUNT = 2 };
int ws_option(const void *ws, int which, int dflt);
void ws_options_forget(const void *ws);
void ws_note_mstep(const void *ws, int code);      // which M-step form the last launch on `ws` took (tests)
// Workgroups that are provably co-resident given the occupancy API's answer for one CU.
int coop_blocks_from_occupancy(int per_cu_api, int block_threads, int cus);
// Number of co-resident workgroups of `kernel` (block threads, dynamic LDS bytes) on this device.
template <class K>
inline int coop_blocks(K kernel, int block_threads, size_t dyn_lds) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dyn_lds) != hipSuccess)
        per_cu = 0;
    return coop_blocks_from_occupancy(per_cu, block_threads, device_info().cus);
}
// Same, cached per (kernel, device): one occupancy query per kernel instantiation and device.
int coop_cap_cached(const void *kernel, int block_threads, size_t dyn_lds);
template <class K>
inline int coop_cap(K kernel, int block_threads, size_t dyn_lds = 0) {
    return coop_cap_cached(reinterpret_cast<const void *>(kernel), block_threads, dyn_lds);
}
// Launch with the launch's OWN return code (hipLaunchKernel), not the process-wide sticky error:
// a stale error of somebody else's call is neither reported as ours nor cleared for its owner.
template <typename... KArgs, size_t... I>
inline int launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                       std::tuple<KArgs...> &vals, std::index_sequence<I...>) {
    void *ptrs[] = {static_cast<void *>(&std::get<I>(vals))..., nullptr};
    return (int)hipLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, lds, st);
}
template <typename... KArgs, typename... Args>
inline int launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                  Args... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};
    return launch_impl(kernel, grid, block, lds, st, vals, std::index_sequence_for<KArgs...>{});
}

// ---------------------------------------------------------------------------------------
// Workspace layout (bytes).  Control words first, each group on its own 256-B line.
// ---------------------------------------------------------------------------------------
struct WsHeader {
    int32_t status;          // sticky RLVI_ST_* flags
    uint32_t pad0a;
    unsigned long long spin_ticks;   // bound of every inter-workgroup wait, 100 MHz ticks (0: default)
    int32_t wls_minnorm;             // raised by the Cholesky kernel for the minimum-norm kernel behind it
    uint32_t pad0[59];
    uint32_t mstep_ticket;   // last-block-done counter of the M-step kernel (self-resetting)
    uint32_t pad1[63];
    uint32_t epoch_base;     // tag base of the exchange slots (advanced by every coop kernel)
    uint32_t pad2[63];
    // self-timed reads-then-writes hold of the M-step (mstep.hip): four slots {shape key, (time bucket << 24) | ticks}
    // -- how long the LAST launch of that shape took to get its tile loads issued, chip-wide; the next launch holds
    // its stores that long
    unsigned long long mstep_hold[4][2];
    uint32_t pad3[48];
};
constexpr size_t WS_HDR_BYTES = sizeof(WsHeader);                       // 1024
static_assert(sizeof(WsHeader) == 1024, "the header ends where the exchange slots begin");
constexpr size_t WS_XCHG_OFF = 1024;
constexpr int XCHG_GRANULES = 4;                                        // per workgroup
// every record is published in XCHG_REPLICAS copies and a workgroup polls copy (blockIdx % 8): 256
// pollers on the same lines serialise at the memory side
constexpr int XCHG_REPLICAS = 8;
constexpr size_t WS_XCHG_BYTES = 2ull * XCHG_REPLICAS * MAX_COOP_WG * XCHG_GRANULES * 8; // 128 KiB
// second exchange region: 8-granule records (three doubles) of the trajectory E-step
constexpr int XCHG2_GRANULES = 8;
constexpr size_t WS_XCHG2_OFF = WS_XCHG_OFF + WS_XCHG_BYTES;
constexpr size_t WS_XCHG2_BYTES = 2ull * MAX_COOP_WG * XCHG2_GRANULES * 8;     // 32 KiB
// warm-start state of the trajectory E-step: {int64 n, int32 k, int32 pad, float nodes[64]}
constexpr size_t WS_TRAJ_OFF = WS_XCHG2_OFF + WS_XCHG2_BYTES;
constexpr size_t WS_TRAJ_BYTES = 512;
constexpr size_t WS_PART_OFF = WS_TRAJ_OFF + WS_TRAJ_BYTES;
constexpr int MSTEP_MAX_BLOCKS = 1024;   // partial records (and so workgroups) of one M-step launch
constexpr int PART_STRIDE = 4;   // per block: {sum pi*l * inv_scale, hits*100/B, sum pi*l, hits}
constexpr size_t WS_PART_BYTES = (size_t)MSTEP_MAX_BLOCKS * PART_STRIDE * 8;
// third exchange region (trajectory E-step, estep_trajb.hip), 64-byte records of eight self-tagged
// fp32 granules {S, P, Q, D, min, R3, R4, -}: stage A [2 parities][64 nodes][256 workgroups], stage B [2][64 nodes]
constexpr int XCHG3_GRANULES = 8;
// a second set of records for calls WITH `out` (their own finalize launch reduces and clears them): such a call may
// be interleaved with an accumulate sequence on the same workspace without touching its records
constexpr size_t WS_PART2_OFF = WS_PART_OFF + WS_PART_BYTES;
constexpr size_t WS_XCHG3A_OFF = WS_PART2_OFF + WS_PART_BYTES;
constexpr size_t WS_XCHG3A_BYTES = 2ull * 64 * MAX_COOP_WG * XCHG3_GRANULES * 8;   // 2 MiB
constexpr size_t WS_XCHG3B_OFF = WS_XCHG3A_OFF + WS_XCHG3A_BYTES;
#ifndef RLVI_XCHG3B_REPLICAS
#define RLVI_XCHG3B_REPLICAS 8
#endif
constexpr int XCHG3B_REPLICAS = RLVI_XCHG3B_REPLICAS;     // the per-node totals are published in 8 copies (one per 32 pollers)
constexpr size_t WS_XCHG3B_BYTES = 2ull * XCHG3B_REPLICAS * 64 * XCHG3_GRANULES * 8;   // 64 KiB
// fourth exchange region (radix-descent threshold, threshold.hip): 32-byte records of four self-tagged
// granules {count, min key, sum lo, sum hi}; a workgroup publishes up to 2 x 256 of them per exchange
// (this digit's bins and, speculatively, the next digit's): stage A [2 parities][256 workgroups][512
// records], stage B (the totals) [2][8 replicas][512 records]
constexpr int THR_BINS = 256;
constexpr int XCHG4_GRANULES = 4;
constexpr int XCHG4B_REPLICAS = 8;
constexpr size_t WS_XCHG4A_OFF = WS_XCHG3B_OFF + WS_XCHG3B_BYTES;
constexpr size_t WS_XCHG4A_BYTES = 2ull * 2 * THR_BINS * MAX_COOP_WG * XCHG4_GRANULES * 8;    // 8 MiB
constexpr size_t WS_XCHG4B_OFF = WS_XCHG4A_OFF + WS_XCHG4A_BYTES;
constexpr size_t WS_XCHG4B_BYTES = 2ull * XCHG4B_REPLICAS * 2 * THR_BINS * XCHG4_GRANULES * 8;    // 256 KiB
// partial Gram matrices of the weighted-least-squares kernel: 8 workgroups x 64 x 64 doubles
constexpr size_t WS_WLS_OFF = WS_XCHG4B_OFF + WS_XCHG4B_BYTES;
constexpr int WLS_MAX_WG = 8;
constexpr size_t WS_WLS_BYTES = (size_t)WLS_MAX_WG * 64 * 64 * 8;               // 256 KiB
// per-workgroup result records of the one-launch in-batch E+M (fused_em.hip): 8 self-tagged granules
// = the two halves of {sum pi*l * inv_scale, hits*100/B, sum pi*l, hits} in fp64
constexpr size_t WS_FEREC_OFF = WS_WLS_OFF + WS_WLS_BYTES;
constexpr size_t WS_FEREC_BYTES = (size_t)MAX_COOP_WG * 8 * 8;                  // 16 KiB
// sharded E-step over several GPUs (estep_trajb.hip, rlvi_estep_sharded_f32): the table of the ranks'
// inboxes as mapped into THIS process (rlvi_workspace_set_peers) and the round counter of the sharded
// solves, which every rank advances alike
constexpr int MAX_PEERS = 8;
struct PeerTable {
    int32_t world, rank;
    uint32_t dtag;                              // last round tag used (rounds of all sharded solves so far)
    uint32_t pad;
    unsigned long long inbox[MAX_PEERS];        // device address of rank r's inbox; [rank] is the local one
};
constexpr size_t WS_PEER_OFF = WS_FEREC_OFF + WS_FEREC_BYTES;
// behind the table: the warm-start state of the SHARDED solves (the same layout as the region at
// WS_TRAJ_OFF).  Only sharded calls touch it and rlvi_workspace_set_peers zeroes it, so it is identical on
// every rank whatever else a rank did with its workspace -- the ranks must agree on how many nodes a
// round evaluates and on the threshold's guesses.
constexpr size_t WS_PEER_TABLE_BYTES = 256;
constexpr size_t WS_PEER_STATE_OFF = WS_PEER_OFF + WS_PEER_TABLE_BYTES;
constexpr size_t WS_PEER_BYTES = WS_PEER_TABLE_BYTES + WS_TRAJ_BYTES;
static_assert(sizeof(PeerTable) <= WS_PEER_TABLE_BYTES, "peer table must fit its workspace region");
// an inbox (uncached device memory that the peers map through an IPC handle and write over xGMI):
// E-step part [2 parities][64 nodes][MAX_PEERS source ranks][8 self-tagged granules] = 64 KiB, then the
// threshold part [2 parities][512 records][MAX_PEERS source ranks][4 granules] = 256 KiB
constexpr size_t PEER_ESTEP_BYTES = 2ull * 64 * MAX_PEERS * 8 * 8;
constexpr size_t PEER_THR_BYTES = 2ull * 512 * MAX_PEERS * 4 * 8;
constexpr size_t PEER_INBOX_BYTES = PEER_ESTEP_BYTES + PEER_THR_BYTES;
constexpr size_t WS_SCRATCH_OFF = WS_PEER_OFF + WS_PEER_BYTES;

__host__ __device__ inline size_t ws_bytes_for(int64_t max_n, int64_t max_b) {
    // scratch: two fp32 vectors of max(max_n, max_b) (fused E+M keeps l and e there)
    int64_t m = max_n > max_b ? max_n : max_b;
    if (m < 0) m = 0;
    size_t s = WS_SCRATCH_OFF + (size_t)m * 8 + 256;
    return (s + 255) & ~(size_t)255;
}

// ---------------------------------------------------------------------------------------
// Cross-lane all-reduce steps without LDS traffic (gfx950):
//   xor 1, 2   DPP quad_perm            xor 4   DPP row_half_mirror (lanes of a quad already agree)
//   xor 8      DPP row_mirror           xor 16  v_permlane16_swap   xor 32  v_permlane32_swap
// Every step pairs two lane sets that already hold identical values, so for a commutative op the
// result is the butterfly all-reduce: all lanes end with bit-identical totals, fixed order.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_x(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ int dpp_x(int v) { return dpp_i<CTRL>(v); }
template <int CTRL>
__device__ __forceinline__ double dpp_x(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)(unsigned)(b & 0xFFFFFFFFll));
    const int hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_x(unsigned long long v) {
    const int lo = dpp_i<CTRL>((int)(unsigned)(v & 0xFFFFFFFFull));
    const int hi = dpp_i<CTRL>((int)(v >> 32));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// {a, b}: a = lanes' own-or-partner value arranged so that op(a, b) is the xor-16 / xor-32 step
template <int W>
__device__ __forceinline__ void swap_pair(int v, int &a, int &b) {
    if (W == 16) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); a = r[0]; b = r[1]; }
    else { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); a = r[0]; b = r[1]; }
}
template <int W, class Op>
__device__ __forceinline__ float swap_step(float v, Op op) {
    int a, b;
    swap_pair<W>(__float_as_int(v), a, b);
    return op(__int_as_float(a), __int_as_float(b));
}
template <int W, class Op>
__device__ __forceinline__ int swap_step(int v, Op op) {
    int a, b;
    swap_pair<W>(v, a, b);
    return op(a, b);
}
template <int W, class Op>
__device__ __forceinline__ double swap_step(double v, Op op) {
    const long long bits = __double_as_longlong(v);
    int alo, blo, ahi, bhi;
    swap_pair<W>((int)(unsigned)(bits & 0xFFFFFFFFll), alo, blo);
    swap_pair<W>((int)(bits >> 32), ahi, bhi);
    return op(__longlong_as_double(((long long)ahi << 32) | (unsigned)alo),
              __longlong_as_double(((long long)bhi << 32) | (unsigned)blo));
}
template <int W, class Op>
__device__ __forceinline__ unsigned long long swap_step(unsigned long long v, Op op) {
    int alo, blo, ahi, bhi;
    swap_pair<W>((int)(unsigned)(v & 0xFFFFFFFFull), alo, blo);
    swap_pair<W>((int)(v >> 32), ahi, bhi);
    return op(((unsigned long long)(unsigned)ahi << 32) | (unsigned)alo,
              ((unsigned long long)(unsigned)bhi << 32) | (unsigned)blo);
}

// All-reduce inside groups of G consecutive lanes (G a power of two, 1..64).
template <int G, typename T, class Op>
__device__ __forceinline__ T group_allreduce(T v, Op op) {
    if (G >= 2) v = op(v, dpp_x<0xB1>(v));     // quad_perm [1,0,3,2]
    if (G >= 4) v = op(v, dpp_x<0x4E>(v));     // quad_perm [2,3,0,1]
    if (G >= 8) v = op(v, dpp_x<0x141>(v));    // row_half_mirror
    if (G >= 16) v = op(v, dpp_x<0x140>(v));   // row_mirror
    if (G >= 32) v = swap_step<16>(v, op);
    if (G >= 64) v = swap_step<32>(v, op);
    return v;
}

struct FAdd { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct FMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return b > a ? b : a; } };
struct FMin { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return b < a ? b : a; } };

template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return group_allreduce<WAVE>(v, FAdd()); }
template <typename T>
__device__ __forceinline__ T wave_max(T v) { return group_allreduce<WAVE>(v, FMax()); }
template <typename T>
__device__ __forceinline__ T wave_min(T v) { return group_allreduce<WAVE>(v, FMin()); }

template <int G>
__device__ __forceinline__ float group_max(float v) { return group_allreduce<G>(v, FMax()); }
template <int G>
__device__ __forceinline__ float group_sum(float v) { return group_allreduce<G>(v, FAdd()); }
template <int G>
__device__ __forceinline__ int group_min_i(int v) { return group_allreduce<G>(v, FMin()); }

// exp(d) for d = z - max <= 0: one multiply by log2(e) and v_exp_f32.  d is an exact-to-1-ulp
// fp32 difference, so the argument error is |d|*log2(e)*2^-24: below 1e-7 relative for every
// term that is not already negligible against sum >= 1 (set RLVI_MSTEP_FAST_EXP=0 for ocml expf).
#ifndef RLVI_MSTEP_FAST_EXP
#define RLVI_MSTEP_FAST_EXP 1
#endif
__device__ __forceinline__ float mexp(float x) {
#if RLVI_MSTEP_FAST_EXP
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#else
    return expf(x);
#endif
}

// Sums the per-block partial records in a fixed order and writes the four output scalars
// (scaled by `scale`: 1 for a single batch, 1/batches for an epoch); optionally clears them.
__device__ __forceinline__ void reduce_partials(double *__restrict__ part, int nblocks, double scale,
                                                float *__restrict__ out, bool clear, int nthreads) {
    double a[PART_STRIDE] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblocks; i += nthreads) {
#pragma unroll
        for (int c = 0; c < PART_STRIDE; ++c) {
            a[c] += part[(size_t)PART_STRIDE * i + c];
            if (clear) part[(size_t)PART_STRIDE * i + c] = 0.0;
        }
    }
    __shared__ double sh[16 * PART_STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < PART_STRIDE; ++c) {
        a[c] = wave_sum(a[c]);
        if (lane == 0) sh[wave * PART_STRIDE + c] = a[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[PART_STRIDE] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < nthreads / 64; ++w)
#pragma unroll
            for (int c = 0; c < PART_STRIDE; ++c) t[c] += sh[w * PART_STRIDE + c];
        out[0] = (float)(t[0] * scale);
        out[1] = (float)(t[1] * scale);
        out[2] = (float)t[2];
        out[3] = (float)t[3];
    }
}

// Order-preserving key of an fp32 value (ascending value <=> ascending unsigned key).  Every NaN,
// whatever its sign bit (0xFFC00000 is the x86 default and what inf - inf gives), takes the
// largest key: NaN orders last, as numpy's argsort and torch.sort place it.
__device__ __forceinline__ uint32_t f32_key(float f) {
    uint32_t b = __float_as_uint(f);
    if (f != f) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __uint_as_float((uint32_t)h << 16);
}
// round-to-nearest-even, NaN kept NaN: the plain cast, which hipcc lowers to v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
// two values in one v_cvt_pk_bf16_f32: lo in bits 15:0, hi in bits 31:16
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

}  // namespace rlvi
