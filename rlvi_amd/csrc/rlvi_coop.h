// Cooperative (multi-workgroup) all-reduce of two doubles inside one launch.
//
// The E-step fixed point is a serial chain of population-wide reductions.  Kernel boundaries
// cost ~1.5-1.9 us each and a software grid barrier ~4 us (MI355X_MICROARCH.md price list), so
// the chain runs inside ONE launch of G <= 256 co-resident workgroups (one per CU) that keep
// their slice of the vector in registers and exchange 32-byte records per step:
//
//   * every workgroup publishes 4 self-tagged 8-byte granules {tag:32, payload:32} (the two
//     doubles, split in halves) with agent-scope relaxed atomic stores (write-through `sc1`);
//   * one wave per workgroup polls all G records with agent-scope relaxed atomic loads until
//     every tag equals the current epoch (the data IS the flag: no fence, no separate flag);
//   * the polled values are combined in a fixed order (ascending workgroup per lane, then a
//     butterfly), so every workgroup computes bit-identical totals and takes identical
//     branches -- the stop decision of the fixed point can never diverge between workgroups.
//
// Slots are double-buffered by step parity: a workgroup can publish step p+2 only after it
// gathered step p+1, i.e. after every workgroup has finished reading step p.  Tags are
// base+step with `base` kept in the workspace and advanced at the end of every launch, so no
// per-launch memset is needed and graph replay is safe.  Every spin is bounded by wall time.
#pragma once
#include "rlvi_common.h"

namespace rlvi {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef unsigned int vu4_t __attribute__((ext_vector_type(4)));

struct OpSum {
    static __device__ __forceinline__ double ident() { return 0.0; }
    static __device__ __forceinline__ double apply(double a, double b) { return a + b; }
};
struct OpMin {
    static __device__ __forceinline__ double ident() { return __builtin_inf(); }
    static __device__ __forceinline__ double apply(double a, double b) { return b < a ? b : a; }
};
struct OpMax {
    static __device__ __forceinline__ double ident() { return -__builtin_inf(); }
    static __device__ __forceinline__ double apply(double a, double b) { return b > a ? b : a; }
};

template <class Op>
struct OpFn {
    __device__ __forceinline__ double operator()(double a, double b) const { return Op::apply(a, b); }
};
template <class Op>
struct OpFnF {
    __device__ __forceinline__ float operator()(float a, float b) const {
        return (float)Op::apply((double)a, (double)b);
    }
};
template <>
struct OpFnF<OpSum> {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
    return group_allreduce<WAVE>(v, OpFn<Op>());
}

// Default bound of an inter-workgroup wait: 100 ms of the 100 MHz wall clock (an exchange takes
// ~1 us; the bound only ends a launch whose workgroups cannot all be resident, e.g. beside another
// process's kernels).  rlvi_workspace_init writes RLVI_SPIN_BOUND_MS into the workspace header.
constexpr unsigned long long SPIN_BOUND_DEFAULT_TICKS = 10000000ull;
__device__ __forceinline__ unsigned long long spin_bound(const WsHeader *hdr) {
    const unsigned long long t = hdr->spin_ticks;
    return t != 0ull ? t : SPIN_BOUND_DEFAULT_TICKS;
}

template <int BLOCK>
struct Coop {
    gu64 *slots;        // [2][XCHG_REPLICAS][MAX_COOP_WG][XCHG_GRANULES]
    int32_t *status;
    uint32_t tag;       // tag of the NEXT exchange
    int step;           // exchanges done so far
    int nwg;
    unsigned long long bound;   // spin bound, wall-clock ticks
    bool dead;          // a wait timed out: stop exchanging, results are invalid

    static constexpr int NW = BLOCK / WAVE;

    // nwg_ = number of exchanging workgroups (blocks 0..nwg_-1 of the grid)
    __device__ __forceinline__ void init(void *ws, int nwg_) {
        char *base = static_cast<char *>(ws);
        WsHeader *hdr = reinterpret_cast<WsHeader *>(base);
        slots = (gu64 *)(reinterpret_cast<unsigned long long *>(base + WS_XCHG_OFF));
        status = &hdr->status;
        // every workgroup reads the base before any workgroup can finish (finishing needs
        // everybody's first publish), so the writer at the end never races this read
        tag = __hip_atomic_load((gu32 *)&hdr->epoch_base, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT) + 1u;
        step = 0;
        nwg = nwg_;
        bound = spin_bound(hdr);
        dead = false;
    }

    // Leaves base + steps in the workspace for the next launch (call from ONE thread, at the end).
    __device__ __forceinline__ void finish(void *ws) {
        WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);
        __hip_atomic_store((gu32 *)&hdr->epoch_base, tag + 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }

    // All threads call with their per-thread partials; all threads return the global result.
    // F32WAVE: the per-thread partials are fp32 values; reduce them inside the wave in fp32 (one
    // fused v_add_f32_dpp per butterfly step instead of two DPP moves + an fp64 add), fp64 from
    // the wave partials on.
    template <class OpA, class OpB, bool F32WAVE = false>
    __device__ __forceinline__ void allreduce2(double &a, double &b) {
        constexpr int NPW = MAX_COOP_WG / WAVE;      // polling waves: one record per lane
        static_assert(NW >= NPW, "the gather needs four waves");
        __shared__ double part[2 * NW];
        __shared__ double part2[2 * NPW];
        __shared__ double bc[2];
        __shared__ int sh_dead;
        const int lane = threadIdx.x & (WAVE - 1);
        const int wave = threadIdx.x / WAVE;
        if (F32WAVE) {
            a = (double)group_allreduce<WAVE>((float)a, OpFnF<OpA>());
            b = (double)group_allreduce<WAVE>((float)b, OpFnF<OpB>());
        } else {
            a = wave_reduce<OpA>(a);
            b = wave_reduce<OpB>(b);
        }
        if (lane == 0) { part[2 * wave] = a; part[2 * wave + 1] = b; }
        if (threadIdx.x == 0) sh_dead = dead ? 1 : 0;
        __syncthreads();
        const bool xchg = nwg > 1 && !dead;          // uniform over the workgroup
        gu64 *buf = slots + (size_t)(step & 1) * XCHG_REPLICAS * MAX_COOP_WG * XCHG_GRANULES;
        if (wave == 0) {
            double ta = lane < NW ? part[2 * lane] : OpA::ident();
            double tb = lane < NW ? part[2 * lane + 1] : OpB::ident();
            ta = wave_reduce<OpA>(ta);
            tb = wave_reduce<OpB>(tb);
            if (xchg) {
                // publish: lane l stores granule l & 3 (32 contiguous bytes) of replica l >> 2
                if (lane < XCHG_GRANULES * XCHG_REPLICAS) {
                    const int gq = lane & (XCHG_GRANULES - 1), rep = lane / XCHG_GRANULES;
                    const unsigned long long bits =
                        (unsigned long long)__double_as_longlong(gq < 2 ? ta : tb);
                    const uint32_t half = (gq & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
                    __hip_atomic_store(buf + ((size_t)rep * MAX_COOP_WG + blockIdx.x) * XCHG_GRANULES + gq,
                                       ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (lane == 0) {
                bc[0] = ta;
                bc[1] = tb;
            }
        }
        if (xchg) {
            // gather: waves 0..3, lane l of wave v owns workgroup 64 v + l (all records in flight
            // together); wave totals, then the four of them in a fixed order
            if (wave < NPW) {
                const int w = wave * WAVE + lane;
                const bool mine = w < nwg;
                gu64 *p = buf + ((size_t)(blockIdx.x & (XCHG_REPLICAS - 1)) * MAX_COOP_WG + (mine ? w : 0)) *
                                XCHG_GRANULES;
                unsigned long long x0 = 0, x1 = 0, x2 = 0, x3 = 0;
                const unsigned long long t0 = wall_clock64();
                bool timeout = false;
                for (unsigned spin = 0;; ++spin) {
                    bool ok = true;
                    if (mine) {
                        // the 32-byte record as two 16-byte sc1 loads (granule-atomic is enough:
                        // every 8-byte granule carries its own tag)
                        vu4_t q0, q1;
                        asm volatile(
                            "global_load_dwordx4 %0, %2, off sc1\n\t"
                            "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                            "s_waitcnt vmcnt(0)"
                            : "=&v"(q0), "=&v"(q1)
                            : "v"((unsigned long long)(uintptr_t)p)
                            : "memory");
                        x0 = ((unsigned long long)q0.y << 32) | q0.x;
                        x1 = ((unsigned long long)q0.w << 32) | q0.z;
                        x2 = ((unsigned long long)q1.y << 32) | q1.x;
                        x3 = ((unsigned long long)q1.w << 32) | q1.z;
                        ok = (uint32_t)(x0 >> 32) == tag && (uint32_t)(x1 >> 32) == tag &&
                             (uint32_t)(x2 >> 32) == tag && (uint32_t)(x3 >> 32) == tag;
                    }
                    if (__all(ok)) break;
                    // the wall clock is read only every 64 polls: keep the poll loop tight
                    if ((spin & 63u) == 63u && wall_clock64() - t0 > bound) {
                        timeout = true;
                        break;
                    }
                }
                double ga = OpA::ident(), gb = OpB::ident();
                if (mine && !timeout) {
                    ga = __longlong_as_double(
                        (long long)(((x1 & 0xFFFFFFFFull) << 32) | (x0 & 0xFFFFFFFFull)));
                    gb = __longlong_as_double(
                        (long long)(((x3 & 0xFFFFFFFFull) << 32) | (x2 & 0xFFFFFFFFull)));
                }
                ga = wave_reduce<OpA>(ga);
                gb = wave_reduce<OpB>(gb);
                if (lane == 0) {
                    part2[2 * wave] = ga;
                    part2[2 * wave + 1] = gb;
                    if (timeout) {
                        sh_dead = 1;
                        atomicOr(status, RLVI_ST_TIMEOUT);
                    }
                }
            }
            __syncthreads();
            a = part2[0];
            b = part2[1];
#pragma unroll
            for (int v = 1; v < NPW; ++v) {
                a = OpA::apply(a, part2[2 * v]);
                b = OpB::apply(b, part2[2 * v + 1]);
            }
        } else {
            __syncthreads();
            a = bc[0];
            b = bc[1];
        }
        dead = sh_dead != 0;
        ++step;
        ++tag;
    }
};

}  // namespace rlvi
