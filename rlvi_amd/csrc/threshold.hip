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

struct Red3 { double s; unsigned long long a, b; };

// Block-wide reduce of {sum (fp64), min (u64), max (u64)}; every thread gets the result.
template <int BLOCK>
__device__ __forceinline__ Red3 block_reduce3(double s, unsigned long long mn,
                                              unsigned long long mx) {
    constexpr int THR_NW = BLOCK / WAVE;
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
template <int E, int THR_BLOCK>
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

// The generic form: any fp32 values (negative, > 1, NaN ordered last), fp64 prefix sums, a bisection
// on the key -- one workgroup of BLOCK threads.  The product path for N > 2 097 152 and the in-kernel
// hand-over of the radix descent when a weight lies outside [0, 1].
template <int E, bool TRUNC, int BLOCK>
__device__ __forceinline__ void threshold_generic(float *__restrict__ w, int64_t N, float alpha,
                                                  float *__restrict__ thr_io, uint8_t *__restrict__ mask,
                                                  int64_t *__restrict__ kept_out) {
    Keys<E, BLOCK> keys;
    keys.load(w, N);

    // beta = alpha * sum(1 - w)   (train_rlvi.py:43-44); global min / max key
    double tot = 0.0;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    keys.for_each([&](uint32_t k, int64_t) {
        tot += one_minus(k);
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    });
    Red3 r = block_reduce3<BLOCK>(tot, kmin, kmax);
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
        r = block_reduce3<BLOCK>(s, 0ull, 0ull);
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
    r = block_reduce3<BLOCK>(s_in, above, below);
    const double S = r.s;
    above = r.a;
    below = r.b;
    r = block_reduce3<BLOCK>(n_in, 0ull, 0ull);
    const double cnt_in = r.s;

    float thr;
    if (below == 0ull) {
        // every element is inside: count = N, threshold = smallest weight
        thr = key_f32((uint32_t)kmin);
    } else {
        const uint32_t v = (uint32_t)(below - 1ull);
        double mult = 0.0;
        keys.for_each([&](uint32_t k, int64_t) { mult += (k == v) ? 1.0 : 0.0; });
        r = block_reduce3<BLOCK>(mult, 0ull, 0ull);
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
        r = block_reduce3<BLOCK>(kept, 0ull, 0ull);
        if (threadIdx.x == 0 && kept_out != nullptr) *kept_out = (int64_t)r.s;
    }
}

template <int E, bool TRUNC>
__global__ __launch_bounds__(THR_BLOCK) void threshold_kernel(float *__restrict__ w, int64_t N,
                                                              float alpha,
                                                              float *__restrict__ thr_io,
                                                              uint8_t *__restrict__ mask,
                                                              int64_t *__restrict__ kept_out) {
    threshold_generic<E, TRUNC, THR_BLOCK>(w, N, alpha, thr_io, mask, kept_out);
}

// ---------------------------------------------------------------------------------------
// Radix descent for weights in [0, 1] (what the E-step produces): the product path.
//
// (1 - pi) * 2^24 is an exact integer <= 2^24, so every prefix sum is a plain integer sum and the
// reference's predicate fl32(prefix) <= beta is evaluated on exactly its numbers.  For pi >= 0 the
// fp32 bit pattern is an order-preserving key.  c* -- the smallest key c such that the keys >= c
// fit together -- is found 8 bits per pass, most significant first: every workgroup builds a
// 256-bin histogram {count, sum of (1-pi) units, smallest key} of its keys inside the current
// prefix, ONE two-stage record exchange (the protocol of estep_trajb.hip: workgroup w publishes a
// record per bin, workgroup b adds bin b's records in a fixed order and publishes the total in 8
// replicas, everybody reads the 256 totals) gives every workgroup identical totals, a suffix scan
// finds the first bin whose suffix fits, and the descent continues in the bin below it.  Four
// exchanges instead of the ~20 dependent ones of a bisection (71 us at N = 65 536), the same exact
// arithmetic, the same bits.  After the last pass the bin below IS the reference's "next key
// below" (v, with its multiplicity), and the smallest key above comes from the min field.
// Outside [0, 1] (NaN, -0.0, negative, > 1): every workgroup leaves the descent at its first exchange
// and workgroup 0 runs the generic fp64 form above on the whole vector, in the same launch (round 1
// enqueued the generic kernel behind this one on a flag: a kernel boundary, 2 us, on every call for a
// case that never happens to an E-step's output).  G <= 256 workgroups of 256 threads, all provably co-resident (launcher);
// G == 1 needs no exchange at all.
// ---------------------------------------------------------------------------------------
constexpr int THQ_BLOCK = 256;
constexpr unsigned long long THQ_ONE = 1ull << 40;      // one key in the packed {count, sum} LDS word
constexpr int THQ_NW = THQ_BLOCK / WAVE;
typedef unsigned int thq_vu4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t one_minus_u(float p) {
    return (uint32_t)__float2uint_rn((1.0f - p) * 16777216.0f);
}

// Wave totals in 32-bit steps.  A 64-bit butterfly step is two DPP moves and a carry chain (or a 64-bit
// compare and two selects); these reductions run on waves that have a SIMD to themselves, where every
// instruction costs its ~6 clocks: counts (< 2^31) and keys (<= 0x3F800000, 0xFFFFFFFF = none) are 32-bit
// values, and a sum below 2^40 is two 32-bit sums of its low 20 bits and the rest.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)wave_sum((int)v); }
__device__ __forceinline__ uint32_t wave_min_key(uint32_t v) {
    const int r = wave_min((int)(v > 0x7FFFFFFFu ? 0x7FFFFFFFu : v));
    return r == 0x7FFFFFFF ? 0xFFFFFFFFu : (uint32_t)r;
}
__device__ __forceinline__ unsigned long long wave_sum_u40(unsigned long long v) {
    const uint32_t lo = wave_sum_u32((uint32_t)v & 0xFFFFFu), hi = wave_sum_u32((uint32_t)(v >> 20));
    return ((unsigned long long)hi << 20) + lo;
}

struct ThqHist {                       // one histogram: what a workgroup publishes / what comes back
    unsigned long long sum[THR_BINS];
    uint32_t cnt[THR_BINS];
    uint32_t mn[THR_BINS];
};

// what a call leaves for the next one (behind the E-step's warm-start state in the workspace)
struct ThrState {
    long long n;                       // population of the last call
    uint32_t key;                      // its v: the largest key whose inclusion did not fit
    uint32_t valid;
};
constexpr size_t WS_THRSTATE_OFF = WS_TRAJ_OFF + 384;
static_assert(384 + sizeof(ThrState) <= WS_TRAJ_BYTES, "threshold state must fit behind the trajectory state");

// Key-list finish: once few keys are left inside the prefix, every workgroup publishes ITS keys (at most 16)
// instead of another pair of 256-bin histograms, everybody reads everybody's, and the remaining digits are
// settled locally on the complete list -- no further exchange, no guess needed.
constexpr int THQ_LIST_CAP = 16;       // keys per workgroup (8 tagged 16-byte pairs); more: the histogram path
constexpr uint32_t THQ_LIST_NONE = 0xFFFFFFFFu, THQ_LIST_OVER = 0xFFFFFFFEu;
constexpr int THQ_LIST_ALL = 2048;     // keys of the complete list kept in LDS (the launcher's list_max stays below)

struct ThqShared {
    ThqHist h[2];                      // [0] this digit, [1] the next digit inside the guessed bin; then the totals
    unsigned long long wsum[2 * THQ_NW];   // cross-wave scratch of the reductions
    unsigned long long wcnt[2 * THQ_NW];
    uint32_t wmin[2 * THQ_NW];
    struct Scan {                      // results of a scan (one slot per histogram: descend(0) and descend(1)
        unsigned long long sum;        //  follow each other without a barrier in between)
        uint32_t beta, b1, cab, mab, g0, below;
    } r[2];
    int dead;
    uint32_t lcount, lover, ltotal;    // key-list finish: this workgroup's keys inside the prefix / somebody's overflow / all keys
    uint32_t lkeys[THQ_LIST_CAP];
    uint32_t lall[THQ_LIST_ALL];       // the complete list (every workgroup's keys inside the prefix), dense
    unsigned long long stamps[60];     // RLVI_THR_DEBUG only
};

// 32-byte record = four granules {tag32 | payload32}: {count, min key, sum lo, sum hi}
__device__ __forceinline__ void thq_store(gu64 *p, uint32_t tag, uint32_t cnt, uint32_t mn,
                                          unsigned long long sum) {
    // two 16-byte write-through stores; every 8-byte granule carries its own tag, so tearing
    // between granules is harmless
    const thq_vu4 q0 = {cnt, tag, mn, tag};
    const thq_vu4 q1 = {(uint32_t)sum, tag, (uint32_t)(sum >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                 "s_nop 1"
                 :
                 : "v"((unsigned long long)(uintptr_t)p), "v"(q0), "v"(q1)
                 : "memory");
}
__device__ __forceinline__ bool thq_load(gu64 *p, uint32_t tag, uint32_t &cnt, uint32_t &mn,
                                         unsigned long long &sum) {
    thq_vu4 q0, q1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1)
                 : "v"((unsigned long long)(uintptr_t)p)
                 : "memory");
    cnt = q0.x; mn = q0.z;
    sum = ((unsigned long long)q1.z << 32) | q1.x;
    return q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag;
}

// Totals of the per-workgroup histograms sh.h[0 .. NH-1] over the G workgroups, back into sh.h
// (identical on every workgroup).  All threads call; returns false after a timeout (sh.dead set).
// Record r = half * 256 + bin lives at slot r of the stage-A row of its workgroup.
// (stamps go to LDS and are copied out at the end: a global store in front of a barrier would cost its round trip)
#define THQ_STAMP() do { if ((RLVI_STAMPS && dbg != nullptr) && blockIdx.x == 0 && threadIdx.x == 0 && dbgi < 60) sh.stamps[dbgi++] = wall_clock64(); } while (0)
// pt != nullptr (sharded over several GPUs): the bin's publishing wave first pushes this rank's totals
// into every rank's inbox, adds up what all ranks pushed (integers: exact, order-free) and publishes the
// global totals; ptag numbers the sharded exchanges of the group (rlvi_trajb.h uses the same counter).
// COARSE0: histogram [0] holds the three coarse bins of a warm call's first digit (and the out-of-range
// count in bin 255) -- only those four records are published, gathered and read back (the other 252 would be
// 1 MB of empty write-through records per exchange).
__device__ __forceinline__ bool thq_coarse_bin(int bin) { return bin < 3 || bin == THR_BINS - 1; }
template <int NH, bool COARSE0 = false>
__device__ __forceinline__ bool thq_exchange(ThqShared &sh, gu64 *bufA, gu64 *bufB, uint32_t tag,
                                             int xstep, int G, WsHeader *hdr,
                                             unsigned long long spin_ticks, unsigned long long *dbg,
                                             int &dbgi, PeerTable *pt = nullptr, uint32_t ptag = 0u) {
    if (G == 1) return true;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int b = (int)blockIdx.x;
    constexpr int NREC = 2 * THR_BINS;                       // slots per workgroup row / totals row
    gu64 *A = bufA + (size_t)(xstep & 1) * NREC * MAX_COOP_WG * XCHG4_GRANULES;
    gu64 *B = bufB + (size_t)(xstep & 1) * XCHG4B_REPLICAS * NREC * XCHG4_GRANULES;
    // ---- stage A: thread t publishes this workgroup's records of bin t
    // (layout [workgroup][record]: the writer's bytes are contiguous -- whole lines per store
    //  instruction; the one-time scattered access is on the reducers' read side)
    // every store instruction of the workgroup covers 4 KiB of its row without gaps: thread t writes
    // the 16-byte half (c & 1) of record c >> 1, c = j * 256 + t, out of the LDS histogram
#pragma unroll
    for (int j = 0; j < 2 * NH; ++j) {
        const int c = j * THQ_BLOCK + tid;
        const int rec = c >> 1, hf = rec >> 8, bin = rec & (THR_BINS - 1);
        thq_vu4 q;
        if (c & 1) {
            const unsigned long long sm = sh.h[hf].sum[bin];
            q = (thq_vu4){(uint32_t)sm, tag, (uint32_t)(sm >> 32), tag};
        } else {
            q = (thq_vu4){sh.h[hf].cnt[bin], tag, sh.h[hf].mn[bin], tag};
        }
        if (!(COARSE0 && hf == 0) || thq_coarse_bin(bin))
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1"
                     :
                     : "v"((unsigned long long)(uintptr_t)(A + (size_t)b * NREC * XCHG4_GRANULES) + (unsigned long long)c * 16ull),
                       "v"(q)
                     : "memory");
    }
    THQ_STAMP();   // stage A stored
    // ---- stage B: workgroup b adds the records of its 256 / G bins (G is 64, 128 or 256): G / 64
    // waves per bin, lane = publishing workgroup, all bins (and both halves) gathered at once
    {
        const int wpb = G >> 6;                              // waves per bin
        const int bin = b + G * (wave / wpb);                // this wave's bin
        const int wg = (wave % wpb) * WAVE + lane;           // the workgroup whose records this lane reads
        uint32_t c[NH], mn[NH];
        unsigned long long sm[NH];
        bool timeout = false;
        {
            const unsigned long long t0 = wall_clock64();
            bool got[NH];
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) {
                got[hf] = COARSE0 && hf == 0 && !thq_coarse_bin(bin);
                c[hf] = 0u; mn[hf] = 0xFFFFFFFFu; sm[hf] = 0ull;
            }
            for (unsigned spin = 0;; ++spin) {
                bool all = true;
#pragma unroll
                for (int hf = 0; hf < NH; ++hf) {
                    if (!got[hf])
                        got[hf] = thq_load(A + ((size_t)wg * NREC + hf * THR_BINS + bin) * XCHG4_GRANULES, tag,
                                           c[hf], mn[hf], sm[hf]);
                    all = all && got[hf];
                }
                if (all) break;
                if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
            }
        }
        if (timeout) sh.dead = 1;
        THQ_STAMP();   // gathered
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            if (timeout) { c[hf] = 0; mn[hf] = 0xFFFFFFFFu; sm[hf] = 0ull; }
            // (a record's sum: at most 8192 keys of at most 2^24 units)
            const unsigned long long tc = wave_sum_u32(c[hf]);
            const unsigned long long ts = wave_sum_u40(sm[hf]);
            const uint32_t tm = wave_min_key(mn[hf]);
            if (lane == 0) { sh.wcnt[hf * THQ_NW + wave] = tc; sh.wsum[hf * THQ_NW + wave] = ts; sh.wmin[hf * THQ_NW + wave] = tm; }
        }
        __syncthreads();
        // the first wave of a bin publishes its totals: lane l stores granule l & 3 of replica l >> 2
        if (wave % wpb == 0 && lane < XCHG4_GRANULES * XCHG4B_REPLICAS && sh.dead == 0) {
            const int gq = lane & 3, rep = lane >> 2;
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) {
                if (COARSE0 && hf == 0 && !thq_coarse_bin(bin)) continue;      // (nobody reads it)
                unsigned long long tc = 0ull, ts = 0ull;
                uint32_t tm = 0xFFFFFFFFu;
                for (int w = wave; w < wave + wpb; ++w) {    // fixed order (integers: any order gives these bits)
                    tc += sh.wcnt[hf * THQ_NW + w]; ts += sh.wsum[hf * THQ_NW + w];
                    tm = sh.wmin[hf * THQ_NW + w] < tm ? sh.wmin[hf * THQ_NW + w] : tm;
                }
                if (pt != nullptr) {
                    // lane = (rank r = rep, granule gq): push, poll, add (the sum's halves are rejoined first)
                    const int r = rep, pworld = pt->world, prank = pt->rank;
                    const bool mine = r < pworld;
                    const uint32_t mv = gq == 0 ? (uint32_t)tc : gq == 1 ? tm : gq == 2 ? (uint32_t)ts
                                                                                : (uint32_t)(ts >> 32);
                    const size_t slot = ((size_t)(ptag & 1u) * NREC + hf * THR_BINS + bin) * MAX_PEERS;
                    if (mine) {
                        gu64 *dst = (gu64 *)(uintptr_t)(pt->inbox[r] + PEER_ESTEP_BYTES) + (slot + prank) * 4 + gq;
                        __hip_atomic_store(dst, ((unsigned long long)ptag << 32) | mv, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    gu64 *src = (gu64 *)(uintptr_t)(pt->inbox[prank] + PEER_ESTEP_BYTES) + (slot + (mine ? r : prank)) * 4 + gq;
                    const unsigned long long t0 = wall_clock64();
                    unsigned long long got = 0ull;
                    bool timeout = false;
                    for (unsigned spin = 0;; ++spin) {
                        got = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const bool ok = !mine || (uint32_t)(got >> 32) == ptag;
                        // (lanes 0..31 of this wave: the others have left the branch)
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
                    }
                    if (timeout) sh.dead = 1;
                    const uint32_t pv = (uint32_t)got;
                    const uint32_t hi = (uint32_t)__shfl_down((int)pv, 1, WAVE);          // gq == 2: the hi half sits one lane up
                    unsigned long long acc = !mine ? (gq == 1 ? 0xFFFFFFFFull : 0ull)
                                             : gq == 2 ? (((unsigned long long)hi << 32) | pv) : (unsigned long long)pv;
#pragma unroll
                    for (int m = 4; m < 32; m <<= 1) {
                        const unsigned long long o = __shfl_xor(acc, m, WAVE);
                        acc = gq == 1 ? (o < acc ? o : acc) : acc + o;
                    }
                    // (each lane uses the one its granule needs; the total's hi half comes from the gq == 2
                    //  lane -- shuffled by ALL lanes: a lane switched off in a branch reads as 0)
                    const unsigned long long from_below = __shfl_up(acc, 1, WAVE);
                    tc = acc; tm = (uint32_t)acc; ts = gq == 3 ? from_below : acc;
                }
                const uint32_t v = gq == 0 ? (uint32_t)tc : gq == 1 ? tm : gq == 2 ? (uint32_t)ts
                                                                           : (uint32_t)(ts >> 32);
                if (sh.dead == 0)
                __hip_atomic_store(B + ((size_t)rep * NREC + hf * THR_BINS + bin) * XCHG4_GRANULES + gq,
                                   ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        THQ_STAMP();   // published
    }
    // ---- everybody: the totals (thread t = bin t of every half), replica blockIdx % 8
    {
        __syncthreads();                                     // (sh.dead of this step, w* free again)
        bool timeout = false;
        uint32_t c[NH], mn[NH];
        unsigned long long sm[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) { c[hf] = 0; mn[hf] = 0xFFFFFFFFu; sm[hf] = 0ull; }
        if (sh.dead == 0) {
            const unsigned long long t0 = wall_clock64();
            bool got[NH];
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) got[hf] = COARSE0 && hf == 0 && !thq_coarse_bin(tid);
            for (unsigned spin = 0;; ++spin) {
                bool all = true;
#pragma unroll
                for (int hf = 0; hf < NH; ++hf) {
                    if (!got[hf])
                        got[hf] = thq_load(B + ((size_t)(b & (XCHG4B_REPLICAS - 1)) * NREC + hf * THR_BINS + tid) *
                                                   XCHG4_GRANULES, tag, c[hf], mn[hf], sm[hf]);
                    all = all && got[hf];
                }
                if (all) break;
                if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
            }
        }
        if (timeout) sh.dead = 1;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) { sh.h[hf].cnt[tid] = c[hf]; sh.h[hf].mn[tid] = mn[hf]; sh.h[hf].sum[tid] = sm[hf]; }
    }
    __syncthreads();
    THQ_STAMP();   // totals in
    if (sh.dead != 0) {
        if (tid == 0) atomicOr(&hdr->status, RLVI_ST_TIMEOUT);
        return false;
    }
    return true;
}

// One hop: workgroup w stores its list (slots past its count: NONE; more than the cap: OVER in slot 0) into the
// head of stage-A row space of this step's parity, every workgroup reads all G lists (thread t the 16-byte pairs
// t, t + 256, ...) and appends the keys it finds to the dense list sh.lall[0 .. sh.ltotal).  Returns 0 = timed out,
// 1 = complete, 2 = somebody overflowed (nothing else has changed: the caller goes on with histograms).
__device__ __forceinline__ int thq_list_hop(ThqShared &sh, gu64 *bufA, uint32_t tag, int xstep, int G, WsHeader *hdr,
                                            unsigned long long spin_ticks, unsigned long long *dbg, int &dbgi) {
    const int tid = threadIdx.x, b = (int)blockIdx.x;
    constexpr int NREC = 2 * THR_BINS;
    THQ_STAMP();   // own keys compacted
    gu64 *Lst = bufA + (size_t)(xstep & 1) * NREC * MAX_COOP_WG * XCHG4_GRANULES;
    if (tid < THQ_LIST_CAP / 2) {
        const uint32_t n = sh.lcount;
        uint32_t k0 = 2 * tid < (int)n ? sh.lkeys[2 * tid] : THQ_LIST_NONE;
        const uint32_t k1 = 2 * tid + 1 < (int)n ? sh.lkeys[2 * tid + 1] : THQ_LIST_NONE;
        if (tid == 0 && n > (uint32_t)THQ_LIST_CAP) k0 = THQ_LIST_OVER;
        const thq_vu4 q = {k0, tag, k1, tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1"
                     :
                     : "v"((unsigned long long)(uintptr_t)(Lst + (size_t)b * THQ_LIST_CAP) + (unsigned long long)tid * 16ull), "v"(q)
                     : "memory");
    }
    // 512 / 1024 / 2048 pairs: 2 / 4 / 8 per thread, four loads in flight at a time (a pair past the end re-reads
    // this thread's first one: always valid)
    const int npairs = G * (THQ_LIST_CAP / 2);
    bool timeout = false, over = false;
    const unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int grp = 0; grp < THQ_LIST_CAP / 8; ++grp) {
        if (grp * 4 * THQ_BLOCK >= npairs) break;                    // (uniform)
        unsigned long long addr[4];
        bool want[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pr = (grp * 4 + q) * THQ_BLOCK + tid;
            want[q] = pr < npairs;
            addr[q] = (unsigned long long)(uintptr_t)Lst + (unsigned long long)(want[q] ? pr : tid) * 16ull;
        }
        thq_vu4 v[4];
        for (unsigned spin = 0;; ++spin) {
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                         "global_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %6, off sc1\n\t"
                         "global_load_dwordx4 %3, %7, off sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                         : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3])
                         : "memory");
            bool all = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) all = all && v[q].y == tag && v[q].w == tag;
            if (all) break;
            if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
        }
        if (timeout) break;
        // this thread's keys into the dense list (any order: the histograms add integers)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!want[q]) continue;
            const uint32_t kk[2] = {v[q].x, v[q].z};
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                if (kk[h2] == THQ_LIST_OVER) over = true;
                else if (kk[h2] != THQ_LIST_NONE) {
                    const uint32_t pos = atomicAdd(&sh.ltotal, 1u);
                    if (pos < (uint32_t)THQ_LIST_ALL) sh.lall[pos] = kk[h2]; else over = true;
                }
            }
        }
    }
    THQ_STAMP();   // lists read
    if (timeout) sh.dead = 1;
    if (over) sh.lover = 1u;
    __syncthreads();
    if (sh.dead != 0) {
        if (tid == 0) atomicOr(&hdr->status, RLVI_ST_TIMEOUT);
        return 0;
    }
    return sh.lover != 0u ? 2 : 1;
}

template <int E, bool TRUNC>
__global__ __launch_bounds__(THQ_BLOCK) void threshold_radix_kernel(
    float *__restrict__ w, int64_t N, float alpha, float *__restrict__ thr_io,
    uint8_t *__restrict__ mask, int64_t *__restrict__ kept_out,
    void *ws, unsigned long long *__restrict__ dbg, int use_state, int64_t Nall, PeerTable *__restrict__ pt,
    int list_max) {
    // (sharded over several GPUs: N weights here, Nall over all ranks, pt the peers' inboxes; the threshold,
    //  the kept count and the warm-start key come out the same on every rank; otherwise Nall == N, pt == nullptr)
    __shared__ ThqShared sh;
    int dbgi = 0;
    bool hand_over = false;
    THQ_STAMP();
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    char *wsb = static_cast<char *>(ws);
    WsHeader *hdr = reinterpret_cast<WsHeader *>(wsb);
    gu64 *bufA = (gu64 *)(reinterpret_cast<unsigned long long *>(wsb + WS_XCHG4A_OFF));
    gu64 *bufB = (gu64 *)(reinterpret_cast<unsigned long long *>(wsb + WS_XCHG4B_OFF));
    ThrState *state = reinterpret_cast<ThrState *>(wsb + (pt != nullptr ? WS_PEER_STATE_OFF + 384 : WS_THRSTATE_OFF));
    uint32_t tag = __hip_atomic_load((gu32 *)&hdr->epoch_base, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const unsigned long long spin_ticks = spin_bound(hdr) * (pt != nullptr ? 100ull : 1ull);
    uint32_t ptag = pt != nullptr ? pt->dtag + 1u : 0u;
    int xstep = 0;
    if (tid == 0) sh.dead = 0;
    // (read before the first exchange: workgroup 0 overwrites them after the last one)
    const float prev = TRUNC ? *thr_io : 0.0f;
    const bool warm = use_state != 0 && state->valid != 0u && state->n == (long long)Nall;
    const uint32_t guess = state->key;                   // the last call's v: its bytes are this call's guesses

    const int64_t L = (N + G - 1) / G;
    const int64_t lo_i = (int64_t)b * L < N ? (int64_t)b * L : N;
    const int64_t hi_i = lo_i + L < N ? lo_i + L : N;
    uint32_t k[E];
    bool have[E];
    uint32_t nbad = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = lo_i + tid + (int64_t)j * THQ_BLOCK;
        have[j] = i < hi_i;
        k[j] = have[j] ? __float_as_uint(w[i]) : 0u;
        if (have[j] && k[j] > 0x3F800000u) { ++nbad; have[j] = false; }   // NaN, negative, -0.0, > 1
    }
    THQ_STAMP();   // slice loaded

    unsigned long long S_base = 0ull, cnt_base = 0ull;   // sum / count of the keys above the current range
    uint32_t above = 0xFFFFFFFFu;                        // smallest key above the current range
    uint32_t prefix = 0u;                                // the high bits fixed so far
    uint32_t gmin = 0u, below_cnt = 0u;
    float beta = 0.0f;
    bool all_inside = false, ok = true;
    auto pred = [&](unsigned long long s_int) {          // fl32(prefix sum) <= beta   (:46-47)
        return (float)((double)s_int * (1.0 / 16777216.0)) <= beta;
    };
    // One digit of the descent on the totals in sh.h[hf]: the first bin b1 whose suffix fits, then the
    // sums / counts / smallest key of the bins >= b1 folded into the running state and the bin below
    // b1 appended to the prefix.  Returns b1 (0: the whole range fits).
    auto descend = [&](int hf, bool first, int digit_of_bin1) {
        ThqHist &H = sh.h[hf];
        // wave 0: suffix sums T(t) = S_base + sum of bins >= t (lane l owns bins 4l..4l+3); the
        // predicate is monotone along the bins (T is non-increasing)
        if (wave == 0) {
            unsigned long long t4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t4[q] = H.sum[4 * lane + q];
            t4[2] += t4[3]; t4[1] += t4[2]; t4[0] += t4[1];
            // suffix sum over the 64 lanes without LDS: row_shl DPP steps inside the 16-lane rows
            // (lanes past the row end read 0), then the totals of the higher rows through v_readlane
            unsigned long long sfx = t4[0];
            sfx += dpp_x<0x101>(sfx);
            sfx += dpp_x<0x102>(sfx);
            sfx += dpp_x<0x104>(sfx);
            sfx += dpp_x<0x108>(sfx);
            {
                auto rl = [&](int src) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sfx, src);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sfx >> 32), src);
                    return ((unsigned long long)hi << 32) | lo;
                };
                const unsigned long long r1 = rl(16), r2 = rl(32), r3 = rl(48);
                const int row = lane >> 4;
                sfx += row == 0 ? r1 + r2 + r3 : row == 1 ? r2 + r3 : row == 2 ? r3 : 0ull;
            }
            const unsigned long long higher = sfx - t4[0];        // bins of the lanes above mine
            if (first) {
                // beta = alpha * sum(1 - w)   (train_rlvi.py:43-44): the total is T(0)
                const unsigned long long total =
                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sfx >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sfx);
                beta = (float)((double)total * (1.0 / 16777216.0)) * alpha;
            }
            int ntrue = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                t4[q] += higher;                                  // T(4 lane + q) - S_base
                ntrue += (int)__popcll(__ballot(pred(S_base + t4[q])));
            }
            const int b1w = THR_BINS - ntrue;                     // first bin whose suffix fits (256: none)
            if (lane == (b1w >> 2) || (b1w == THR_BINS && lane == 0))
                sh.r[hf].sum = b1w == THR_BINS ? 0ull : t4[b1w & 3];
            // count / smallest key of the bins >= b1, first: the global minimum -- on this wave's own four bins
            // per lane (round 2: a masked block reduction by all waves, two more barriers per digit)
            uint32_t c4 = 0u, m4 = 0xFFFFFFFFu, g4 = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int bin = 4 * lane + q;
                const uint32_t cq = H.cnt[bin], mq = H.mn[bin];
                if (bin >= b1w) { c4 += cq; m4 = mq < m4 ? mq : m4; }
                g4 = mq < g4 ? mq : g4;
            }
            const uint32_t cab = wave_sum_u32(c4), mab = wave_min_key(m4);
            const uint32_t g0 = first ? wave_min_key(g4) : 0u;
            if (lane == 0) {
                sh.r[hf].b1 = (uint32_t)b1w; sh.r[hf].beta = __float_as_uint(beta);
                sh.r[hf].cab = cab; sh.r[hf].mab = mab; sh.r[hf].g0 = g0;
                sh.r[hf].below = b1w > 0 ? H.cnt[b1w - 1] : 0u;
            }
        }
        __syncthreads();
        const int b1 = (int)sh.r[hf].b1;
        if (first) { beta = __uint_as_float(sh.r[hf].beta); gmin = sh.r[hf].g0; }
        if (b1 > 0) {
            S_base += sh.r[hf].sum;
            cnt_base += (unsigned long long)sh.r[hf].cab;
            above = sh.r[hf].mab < above ? sh.r[hf].mab : above;
            // (coarse first digit: bin 1 stands for the guessed top byte)
            prefix = (prefix << 8) | (uint32_t)(digit_of_bin1 >= 0 ? digit_of_bin1 : b1 - 1);
            below_cnt = sh.r[hf].below;
        }
        // (no barrier behind: the next scan of THIS slot comes after the barriers of a histogram pass)
        return b1;
    };

    // Two digits per exchange when the last call's v is a usable guess: histogram [0] is this digit,
    // [1] the NEXT digit of the keys inside the bin the guess names.  If the descent picks that very
    // bin, its totals are already here and the next exchange is saved (a warm call: two exchanges for
    // the four digits); if not, nothing is lost.
    // A warm call does not histogram the first digit at all (every key takes part and half of them
    // share a top byte: 64 lanes on one LDS word): three coarse bins {top byte below / equal to /
    // above the guessed one}, accumulated in registers and reduced per wave, say whether the guess
    // holds -- the scan must pick bin 1 -- and carry the total (beta), the global minimum and
    // everything about the keys above.  A wrong guess costs that one exchange and restarts cold.
    int level = 0;
    bool use_warm = warm;
    // Key-list finish (thq_list_hop): as soon as at most list_max keys are left inside the prefix (four per
    // workgroup on average: a workgroup with more than 16 is then one in a million), the workgroups publish
    // their keys instead of histograms and everybody settles the remaining digits on the complete list, without
    // another exchange and whatever the last call's key was.  Same integers, same predicate, same bits.
    bool listed = false;
#pragma unroll 1
    while (level < 4 && ok && !all_inside) {
        const int shift = 24 - 8 * level;
        if (!listed && level >= 1 && pt == nullptr && G > 1 && (long long)below_cnt <= (long long)list_max) {
            if (tid == 0) { sh.lcount = 0u; sh.lover = 0u; sh.ltotal = 0u; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (have[j] && (k[j] >> (shift + 8)) == prefix) {
                    const uint32_t pos = atomicAdd(&sh.lcount, 1u);
                    if (pos < (uint32_t)THQ_LIST_CAP) sh.lkeys[pos] = k[j];
                }
            __syncthreads();
            const int r = thq_list_hop(sh, bufA, tag, xstep, G, hdr, spin_ticks, dbg, dbgi);
            ++tag; ++xstep;
            THQ_STAMP();   // lists in
            if (r == 0) { ok = false; break; }
            listed = r == 1;
        }
        // (the guess applies while the prefix fixed so far agrees with its leading bytes)
        const bool spec = !listed && use_warm && level < 3 && (level == 0 || (guess >> (shift + 8)) == prefix);
        const bool coarse = spec && level == 0;
        const uint32_t gbin = (guess >> shift) & 0xFFu;
        // ---- local histograms of the keys inside the current prefix
        sh.h[0].cnt[tid] = 0u; sh.h[0].mn[tid] = 0xFFFFFFFFu; sh.h[0].sum[tid] = 0ull;
        if (spec) { sh.h[1].cnt[tid] = 0u; sh.h[1].mn[tid] = 0xFFFFFFFFu; sh.h[1].sum[tid] = 0ull; }
        __syncthreads();                                  // (the first one also waits for the slowest wave's slice: 0.6 us)
        uint32_t cc[3] = {0u, 0u, 0u}, cm[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        unsigned long long cs[3] = {0ull, 0ull, 0ull};
        auto hist_key = [&](bool hv, uint32_t key) {
            const bool in = hv && (level == 0 || (key >> (shift + 8)) == prefix);
            if (in) {
                const uint32_t bin = (key >> shift) & 0xFFu;
                const unsigned long long om = (unsigned long long)one_minus_u(__uint_as_float(key));
                if (coarse) {
                    const int c3 = bin < gbin ? 0 : (bin == gbin ? 1 : 2);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (q == c3) { cc[q] += 1u; cs[q] += om; cm[q] = key < cm[q] ? key : cm[q]; }
                } else {
                    // (count and sum in ONE LDS atomic: count << 40 | sum; a workgroup holds at most
                    //  8192 keys of at most 2^24 units each)
                    // (measured and not kept, round 3: reducing the keys of a wave that share a bin across the
                    //  wave and adding them by one lane -- a truncated vector's zeros put whole waves into one
                    //  bin -- costs more than the LDS serialising the 64 lanes: 21.6 -> 22.05 us)
                    atomicAdd(&sh.h[0].sum[bin], THQ_ONE | om);
                    atomicMin(&sh.h[0].mn[bin], key);
                }
                if (spec && bin == gbin) {
                    const uint32_t bin2 = (key >> (shift - 8)) & 0xFFu;
                    atomicAdd(&sh.h[1].sum[bin2], THQ_ONE | om);
                    atomicMin(&sh.h[1].mn[bin2], key);
                }
            }
        };
        if (listed) {
            const int nl = (int)sh.ltotal;
            for (int i = tid; i < nl; i += THQ_BLOCK) hist_key(true, sh.lall[i]);
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) hist_key(have[j], k[j]);
        }
        if (coarse) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned long long wc = wave_sum_u32(cc[q]);
                const unsigned long long wsm = wave_sum_u40(cs[q]);       // (a thread: at most 32 keys)
                const uint32_t wm = wave_min_key(cm[q]);
                if (lane == 0) {
                    atomicAdd(&sh.h[0].sum[q], (wc << 40) | wsm);
                    atomicMin(&sh.h[0].mn[q], wm);
                }
            }
        }
        if (level == 0 && nbad != 0u) atomicAdd(&sh.h[0].sum[THR_BINS - 1], (unsigned long long)nbad << 40);   // no valid key has top byte 0xFF
        __syncthreads();
        {   // unpack count << 40 | sum
            const unsigned long long p0 = sh.h[0].sum[tid];
            sh.h[0].cnt[tid] = (uint32_t)(p0 >> 40);
            sh.h[0].sum[tid] = p0 & (THQ_ONE - 1ull);
            if (spec) {
                const unsigned long long p1 = sh.h[1].sum[tid];
                sh.h[1].cnt[tid] = (uint32_t)(p1 >> 40);
                sh.h[1].sum[tid] = p1 & (THQ_ONE - 1ull);
            }
        }
        __syncthreads();
        THQ_STAMP();   // histograms built
        if (!listed) {                                    // (a complete list: the local histogram IS the total)
            ok = coarse ? thq_exchange<2, true>(sh, bufA, bufB, tag, xstep, G, hdr, spin_ticks, dbg, dbgi, pt, ptag)
                 : spec ? thq_exchange<2>(sh, bufA, bufB, tag, xstep, G, hdr, spin_ticks, dbg, dbgi, pt, ptag)
                        : thq_exchange<1>(sh, bufA, bufB, tag, xstep, G, hdr, spin_ticks, dbg, dbgi, pt, ptag);
            ++tag; ++xstep; ++ptag;
        }
        if (!ok) break;
        if (level == 0) {
            // out of [0, 1]: the generic form (workgroup 0, below) takes over
            const bool range_bad = sh.h[0].cnt[THR_BINS - 1] != 0u;
            if (range_bad) { ok = false; hand_over = true; break; }
        }
        const int b1 = descend(0, level == 0, coarse ? (int)gbin : -1);
        if (b1 == 0) {
            // every key of the range fits (only possible at level 0: a deeper range was entered
            // because it does NOT fit as a whole): count = N, threshold = smallest weight
            all_inside = true;
            break;
        }
        if (coarse && b1 != 2) {
            // the guessed top byte is not the one: start over without guesses
            use_warm = false;
            S_base = 0ull; cnt_base = 0ull; above = 0xFFFFFFFFu; prefix = 0u;
            continue;
        }
        ++level;
        if (spec && (coarse || (uint32_t)(b1 - 1) == gbin)) {
            descend(1, false, -1);                       // (b1 >= 1 here: the bin as a whole does not fit)
            ++level;
        }
        THQ_STAMP();   // scan done
    }

    float thr = 0.0f;
    if (ok) {
        if (all_inside) {
            thr = __uint_as_float(gmin);
        } else {
            // prefix is now the full key v = the largest key whose inclusion (all copies) does not
            // fit; S_base / cnt_base are the sum / count of the keys above it
            const uint32_t v = prefix;
            const unsigned long long t = one_minus_u(__uint_as_float(v));
            // how many of v's copies still fit: the boundary of a monotone predicate on (jl, jh), 64 candidates
            // per step -- one per lane, a ballot counts the ones that fit (exact ties and the unvisited slots'
            // zeros come in thousands of copies: two steps instead of a dozen dependent bisections, every
            // wave the same)
            long long jl = 0, jh = (long long)below_cnt;  // pred(S + jl t) true, pred(S + jh t) false
            while (jh - jl > 1) {
                const long long step = (jh - jl + 63) >> 6;
                const long long jm = jl + (long long)(lane + 1) * step;
                const bool fits = jm < jh && pred(S_base + (unsigned long long)jm * t);
                const long long n = (long long)__popcll(__ballot(fits));
                const long long up = jl + (n + 1) * step;
                jl += n * step;
                jh = up < jh ? up : jh;
            }
            const long long j = jl;
            if (cnt_base + (unsigned long long)j == 0ull) thr = __uint_as_float(gmin);   // last_index = -1 wraps (:47-48)
            else if (j >= 1) thr = __uint_as_float(v);
            else thr = __uint_as_float(above);
        }
        if (TRUNC && !(thr > prev)) thr = prev;           // threshold = max(threshold, criterion) (:102)
    }

    THQ_STAMP();   // threshold known
    if (TRUNC && ok) {
        uint32_t kept = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t i = lo_i + tid + (int64_t)j * THQ_BLOCK;
            if (have[j]) {
                float x = __uint_as_float(k[j]);
                if (x < thr) { x = 0.0f; w[i] = 0.0f; }       // :103
                const bool mk = x > thr;                      // main.py:343
                if (mask != nullptr) mask[i] = mk ? 1 : 0;
                kept += mk ? 1u : 0u;
            }
        }
        if (kept_out != nullptr) {                            // uniform: one more exchange, bin 0 carries the count
            sh.h[0].cnt[tid] = 0u; sh.h[0].mn[tid] = 0xFFFFFFFFu; sh.h[0].sum[tid] = 0ull;
            __syncthreads();
            const unsigned long long kw = wave_sum_u32(kept);
            if (lane == 0) atomicAdd(&sh.h[0].sum[0], kw);
            __syncthreads();
            ok = thq_exchange<1>(sh, bufA, bufB, tag, xstep, G, hdr, spin_ticks, dbg, dbgi, pt, ptag);
            ++tag; ++xstep; ++ptag;
            if (ok && b == 0 && tid == 0) *kept_out = (int64_t)sh.h[0].sum[0];
        }
    }
    THQ_STAMP();   // truncated
    if (b == 0 && tid == 0) {
        if (ok) *thr_io = thr;
        // this call's v for the next call's guesses (every workgroup read the old one before its first exchange)
        state->n = (long long)Nall;
        state->key = prefix;
        if (pt != nullptr) pt->dtag = ptag - 1u;
        state->valid = (ok && !all_inside) ? 1u : 0u;
        if ((RLVI_STAMPS && dbg != nullptr)) {
            for (int q = 0; q < dbgi; ++q) dbg[q] = sh.stamps[q];
            dbg[63] = (unsigned long long)dbgi;
        }
        __hip_atomic_store((gu32 *)&hdr->epoch_base, tag + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (hand_over && b == 0) {       // (nobody has written anything: every workgroup left at the first exchange)
        __syncthreads();
        if (pt != nullptr) {
            // sharded: the generic form sees one rank's weights only -- report instead (every rank saw the
            // same totals, so every rank lands here)
            if (tid == 0) atomicOr(&hdr->status, RLVI_ST_NOCONV);
        } else {
            threshold_generic<0, TRUNC, THQ_BLOCK>(w, N, alpha, thr_io, mask, kept_out);
        }
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

// Every N up to 2 097 152: the radix descent on G = min(256, N / 256, what is provably co-resident)
// workgroups (its workgroup 0 runs the generic fp64 form itself when a weight lies outside [0, 1]).
// Larger: the generic form, one workgroup, streaming.
template <bool TRUNC>
static int launch_threshold(float *w, int64_t N, float alpha, float *thr, uint8_t *mask,
                            int64_t *kept, void *ws, hipStream_t st, int64_t n_all = 0, int sharded = 0,
                            int dry_run = 0) {
    // dry_run: only say whether a launch would be admitted for this shape with the co-residency this process is
    // entitled to (rlvi_threshold_sharded_check); nothing is launched, ws may be NULL
    const int64_t Nall = sharded ? n_all : N;
    PeerTable *pt = (sharded && !dry_run) ? reinterpret_cast<PeerTable *>(static_cast<char *>(ws) + WS_PEER_OFF) : nullptr;
    unsigned long long *dbg = (!dry_run && tune_get("RLVI_THR_DEBUG", 0))
                                  ? reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF + 256)
                                  : nullptr;
    int rc = RLVI_E_LIMIT;
    bool launched = false;
    // 0: never use the last call's key as a guess (lab knob; the caller's form: the workspace option "cold_start")
    const int use_state = tune_get("RLVI_THR_WARM", (!sharded && !dry_run && ws_option(ws, WSOPT_COLD_START, 0)) ? 0 : 1);
    // key-list finish once at most this many keys PER WORKGROUP (on average) are left inside the prefix; 0: never
    const int list_per_wg = tune_get("RLVI_THR_LIST", 4);
#define RLVI_THQ(E_, G_)                                                                         \
    do {                                                                                         \
        auto kern = threshold_radix_kernel<E_, TRUNC>;                                           \
        const int g_ = (G_);                                                                     \
        if (!launched && (g_ == 1 || coop_cap(kern, THQ_BLOCK) >= g_) &&                           \
            (N + g_ - 1) / g_ <= (int64_t)(E_) * THQ_BLOCK) {                                      \
            rc = dry_run ? 0                                                                     \
                         : launch(kern, dim3((unsigned)g_), dim3(THQ_BLOCK), 0, st, w, N, alpha, thr, mask, \
                                  kept, ws, dbg, use_state, Nall, pt,                            \
                                  list_per_wg * g_ < THQ_LIST_ALL ? list_per_wg * g_ : THQ_LIST_ALL); \
            launched = true;                                                                     \
        }                                                                                        \
    } while (0)
    // G = 1 (no exchange) up to 1024 keys; 64 workgroups (fewest participants per exchange) while a
    // slice fits 8 keys per thread, then 128, then 256; every G > 1 only if the occupancy query says
    // that many workgroups are co-resident on this device
    if (tune_get("RLVI_THR_RADIX", 1)) {
        const int force_g = tune_get("RLVI_THR_G", 0);
        if (N <= 1024 && !force_g && !sharded) RLVI_THQ(4, 1);
        // the histogram costs about 1 us per key slot of a thread (LDS atomics), an exchange -- and the key
        // list's hop -- grows with the number of publishing workgroups: 64 workgroups up to two keys per
        // thread, 128 up to sixteen, then 256 (round 3, criterion alone: N = 32 768 16.5 / 17.3 us for 64 / 128;
        // 65 536 18.5 / 18.1 / 20.4 for 64 / 128 / 256; 262 144 21.3 / 23.0 and 524 288 24.0 / 24.4 for 128 / 256;
        // 1 048 576 29.0 / 26.7)
        for (int G = 64; G <= 256; G *= 2) {
            if (force_g && G != force_g) continue;
            const int64_t L = (N + G - 1) / G;
            const int emax = force_g ? 32 : (G == 64 ? 2 : G == 128 ? 16 : 32);
            if (L > (int64_t)THQ_BLOCK * emax) continue;
            if (L <= THQ_BLOCK * 1) RLVI_THQ(1, G);
            else if (L <= THQ_BLOCK * 2) RLVI_THQ(2, G);
            else if (L <= THQ_BLOCK * 4) RLVI_THQ(4, G);
            else if (L <= THQ_BLOCK * 8) RLVI_THQ(8, G);
            else if (L <= THQ_BLOCK * 16) RLVI_THQ(16, G);
            else if (L <= THQ_BLOCK * 32) RLVI_THQ(32, G);
        }
        // (fewer co-resident workgroups than the preferred geometry asks for: any geometry that fits)
        for (int G = 64; G <= 256 && !launched && !force_g; G *= 2) {
            const int64_t L = (N + G - 1) / G;
            if (L <= THQ_BLOCK * 8) RLVI_THQ(8, G);
            else if (L <= THQ_BLOCK * 32) RLVI_THQ(32, G);
        }
        if (N <= 8192 && !sharded) RLVI_THQ(32, 1);          // (fewer than 64 co-resident workgroups: one workgroup)
    }
#undef RLVI_THQ
    if (launched) return rc;
    if (sharded) return RLVI_E_LIMIT;            // (the one-workgroup forms see one rank's weights only)
    if (dry_run) return 0;
    return launch(threshold_kernel<0, TRUNC>, dim3(1), dim3(THR_BLOCK), 0, st, w, N, alpha, thr, mask,
                  kept);
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

// Sharded over the GPUs of one node (rlvi_estep_sharded_f32's companion): this rank's n_local weights,
// n_all over all ranks.  *thr_inout = max(*thr_inout, criterion over ALL weights) and *kept_out = the
// number of weights above it over ALL ranks come out identical everywhere; the truncation and the mask
// cover this rank's weights.  A collective; RLVI_E_LIMIT outside 1024 < n_local <= 2 097 152.
namespace rlvi { int peers_world_of(const void *ws); }
extern "C" int rlvi_threshold_truncate_sharded_f32(float *weights, int64_t n_local, int64_t n_all, float alpha,
                                                   float *thr_inout, uint8_t *mask_gt, int64_t *kept_out,
                                                   void *ws, void *stream) {
    if (!weights || !thr_inout || !ws) return RLVI_E_NULL;
    if (n_local <= 0 || n_all < n_local) return RLVI_E_SHAPE;
    if (n_all > (1ll << 29)) return RLVI_E_LIMIT;
    if (((uintptr_t)weights & 3) || ((uintptr_t)thr_inout & 3) || ((uintptr_t)kept_out & 7))
        return RLVI_E_ALIGN;
    if (peers_world_of(ws) < 1) return RLVI_E_WS;
    if (n_local <= 1024) return RLVI_E_LIMIT;
    return launch_threshold<true>(weights, n_local, alpha, thr_inout, mask_gt, kept_out, ws,
                                  static_cast<hipStream_t>(stream), n_all, 1);
}

// Would rlvi_threshold_truncate_sharded_f32 launch for this shape on this device, now?  0 = yes, RLVI_E_LIMIT =
// no; nothing is launched (occupancy queries only).  The companion of rlvi_estep_sharded_check: every rank asks
// both before the first sharded epoch end and the ranks compare answers (rlvi_amd.dist.set_owner_sharding) -- a
// rank that cannot launch would leave the others spinning on its records until their bound.
extern "C" int rlvi_threshold_sharded_check(int64_t n_local, int64_t n_all) {
    if (n_local <= 0 || n_all < n_local) return RLVI_E_SHAPE;
    if (n_all > (1ll << 29) || n_local <= 1024) return RLVI_E_LIMIT;
    return launch_threshold<true>(nullptr, n_local, 0.05f, nullptr, nullptr, nullptr, nullptr, nullptr, n_all, 1, 1);
}

extern "C" int rlvi_truncate_f32(float *weights, int64_t N, const float *thr, uint8_t *mask_gt,
                                 void *stream) {
    if (!weights || !thr) return RLVI_E_NULL;
    if (N <= 0) return RLVI_E_SHAPE;
    int64_t nb = (N + 255) / 256;
    if (nb > 2048) nb = 2048;
    return launch(truncate_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream),
                  weights, N, thr, mask_gt);
}
