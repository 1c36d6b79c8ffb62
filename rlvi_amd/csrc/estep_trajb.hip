// Trajectory E-step (train_rlvi.py:14-38; N from 64 to 2 097 152: the bench size, every dataset
// of the reference, the weak-scaling bench up to 8 x 65 536, and a mini-batch's worth of samples).
//
// Guessed nodes r'_k of the whole fixed-point trajectory (rlvi_traj.h holds the recurrence), per
// round the totals S(r'_k), dS/dr, -d2S/dr2 / 2 and D(r'_k, r'_{k-1}) for all nodes, a corrected
// scalar recurrence, repeat until the nodes stop moving (two rounds, warm or cold).  Each of up to
// 256 workgroups owns a small slice of the samples (N/256, registers) and evaluates ALL live nodes on it:
//   * eight nodes at a time: per-thread partials of {sum r e/(1+r e), sum e/(1+r e)^2, sum e^2/(1+r e)^3,
//     sum d^2}, sample pairs in packed fp32, then a transposing butterfly (v_permlane32_swap /
//     v_permlane16_swap / row_ror:8 halve the number of live values at every step) leaves node
//     (lane>>3)&7 of the chunk in each lane: 19 cross-lane operations per quantity for 8 nodes
//     instead of 48;
//   * only the nodes that matter are evaluated: Ke = (iterations of the previous call) + margin,
//     all of them when no stop index shows up among those;
//   * the exchange has two stages: workgroup w publishes one 48-byte record per node (stage A);
//     workgroup k < Ke gathers node k's 256 records, adds them in a fixed order and publishes the
//     total (stage B); wave 0 of every workgroup reads the Ke totals (lane k = node k) and runs
//     the recurrence.  What a stage costs was decided by two things: a lane's write-through (sc1)
//     stores leave one after the other, so every publish is ONE store per lane (four waves store
//     S, S', Q, D of stage A; the 64 lanes of a wave the 8 x 8 granules of stage B); and 256
//     workgroups polling the same 768 bytes serialise at the memory side, so the totals go out in
//     8 replicas and a workgroup polls replica blockIdx % 8 (2.4-2.9 -> 0.8 us for that hop).
// Pads have e = 0 and drop out of every sum.  Per-thread and per-wave sums are fp32; cross-wave and
// cross-workgroup sums of S are fp64, in a fixed order, so every workgroup sees bit-identical totals
// and identical inputs + workspace state give identical bits.  Protocol (sc1 stores / loads,
// self-tagged granules, parity buffers, tags from the workspace base, wall-clock-bounded spins) as in
// rlvi_coop.h.
#include "rlvi_trajb.h"

namespace rlvi {

// (one-sample slices at 256 threads: three waves per SIMD -- at most 168 registers -- so that the occupancy
//  query proves 512 workgroups co-resident: the full 256-workgroup grid keeps its one-sample slices even when two
//  processes share the device)
// (measured and not kept, round 3: 512-thread workgroups whose second four waves hold no samples and only take
//  their share of the nodes in the sums, as the in-batch kernel's spare waves do -- 22.1 against 21.9 us per step)
#ifndef RLVI_TB_MINW
#define RLVI_TB_MINW 3
#endif
template <int E, int TB_BLOCK>
__global__ __launch_bounds__(TB_BLOCK, (TB_BLOCK == 256 && E == 1 && !RLVI_STAMPS) ? RLVI_TB_MINW : 1) void estep_trajb_kernel(
    float *__restrict__ res, float *__restrict__ wts, int64_t N, float tol, int K,
    int32_t *__restrict__ out_iters, float *__restrict__ trace, void *ws,
    float *__restrict__ mstep_out, double mstep_scale, unsigned long long *__restrict__ dbg, int G,
    int64_t Nall, PeerTable *__restrict__ pt, int verify) {
    // G <= TB_G exchanging workgroups (what is provably co-resident on this device).  The epoch end's
    // reduction of the M-step records (mstep_out != nullptr) is done by workgroup G - 1 inside the solve, in
    // the time it would otherwise wait for the first round's totals -- no extra workgroup (round 2 had one:
    // 257 workgroups on 256 CUs, which is what forced this kernel under 168 registers)
    __shared__ TbShared<TB_BLOCK / WAVE, tb_stage(E, TB_BLOCK)> sh;
    const int tid = threadIdx.x;
#if RLVI_STAMPS
    // (lab build: shader clock of this launch = delta s_memtime / delta s_memrealtime x 100 MHz)
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int b = (int)blockIdx.x;
    const int64_t L = (N + G - 1) / G;
    const int64_t lo = (int64_t)b * L < N ? (int64_t)b * L : N;
    const int64_t hi = lo + L < N ? lo + L : N;
    // (sharded over several GPUs: N samples here, Nall over all ranks, pt the peers' inboxes;
    //  otherwise Nall == N and pt == nullptr)
    // (verify bit 1: the caller asked for a cold start -- the workspace option "cold_start": no guess from the last call)
    const TbWarm wm = tb_warm(ws, Nall, K, pt != nullptr, (verify & 2) != 0);

    // ---- slice -> registers: raw residuals and the caller's pi
    float l[E], ev[E], q0[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = lo + tid + (int64_t)j * TB_BLOCK;
        const bool ok = i < hi;
        l[j] = ok ? res[i] : __builtin_inff();
        q0[j] = ok ? wts[i] : 0.0f;
    }
    const TbSolved s = trajb_solve<E, TB_BLOCK>(sh, wm, l, q0, ev, true, b, G, Nall, tol, K, out_iters, trace,
                                                ws, dbg, pt, (verify & 1) != 0, mstep_out, mstep_scale);
    // a wait that timed out (RLVI_ST_TIMEOUT: the workgroups were not all resident) leaves the
    // caller's residuals and pi as they were -- the host raises on the status; it never hands out garbage
#if RLVI_STAMPS
    if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        dbg[960] = __builtin_amdgcn_s_memtime() - clk0;
        dbg[961] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
    if (s.dead) return;
    const float pmax = tb_pmax(s);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = lo + tid + (int64_t)j * TB_BLOCK;
        if (i < hi) {
            res[i] = l[j] - s.gmin;                     // residuals.sub_(min) (:27)
            wts[i] = tb_weight(s, pmax, ev[j]);         // pi / max(pi) (:38)
        }
    }
}

// Eligibility + launch.  Returns 1 if launched (rc in *rc), 0 if not applicable.  dry_run: only say
// whether a launch would be admitted (rlvi_estep_sharded_check: every rank asks before the first
// collective launch, so that no rank launches a solve that a peer cannot join).
// (Round 1 had a node-per-workgroup form for 4096 <= N < 12 288 -- 240 workgroups of 1024 threads,
//  0.5 us ahead at N = 8192; it needed a whole CU per workgroup to be free and was retired.)
int try_launch_estep_trajb(float *res, float *wts, int64_t N, float tol, int maxiter,
                           int32_t *out_iters, float *trace, void *ws, hipStream_t st,
                           float *mstep_out, double mstep_scale, int *rc, int64_t n_all, int sharded,
                           int dry_run) {
    const int mode = tune_get("RLVI_ESTEP_TRAJB", 1);
    // (round 3: from 64 samples on -- the solve costs its ~10.5 us whatever N is, the iterative kernel 0.8-1.8 us
    //  per iteration: N = 256 16.5 -> 12.0 us, 1024 20.6 -> 12.3, 3000 37.9 -> 12.4 with the copy kernel of the
    //  timing loop; rounds 1-2 had started at 4096 because the slices of their node-per-workgroup form needed it)
    const int64_t nmin = tune_get("RLVI_ESTEP_TRAJB_NMIN", 64);
    if (mode == 0 || maxiter < 1 || maxiter > TJ_MAXK || N < nmin) return 0;
    const int debug = tune_get("RLVI_TJ_DEBUG", 0);
    // 256 threads measured 2-3 us per call ahead up to N = 262 144, level at 524 288, 1 us behind
    // from 1e6 on (whole step / eager call, hipGraph): 512 threads only for slices beyond 4096
    const int blk = tune_get("RLVI_TB_BLOCK", 0);
    unsigned long long *dbg = (debug && !dry_run) ? reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF) : nullptr;
    const int64_t Nall = sharded ? n_all : N;
    PeerTable *pt = (sharded && !dry_run) ? reinterpret_cast<PeerTable *>(static_cast<char *>(ws) + WS_PEER_OFF) : nullptr;
    // bit 0: always run the verification round (lab knob); bit 1: ignore the previous call's trajectory (the
    // caller's workspace option "cold_start": every call as the reference's loop starts it, train_rlvi.py:29)
    const int verify = (tune_get("RLVI_TJ_VERIFY", 0) ? 1 : 0) |
                       ((!sharded && !dry_run && ws_option(ws, WSOPT_COLD_START, 0)) ? 2 : 0);
    int launched = 0;
    // The exchanging workgroups wait for each other, so all of them (and the reduction workgroup) must
    // be resident at once: a geometry (E samples per thread, B threads) runs on G = min(TB_G, what
    // coop_cap proves co-resident for THIS instantiation on this device -- 1/S of it when S processes
    // share the device) workgroups and is admitted when G >= TJ_MAXK
    // (node k is reduced by workgroup k) and the slice N/G fits E x B.  The candidates are tried in the
    // order of their slice length, so with the whole device available the choice is the one measured
    // fastest (the smallest slice that holds N/256), and with fewer co-resident workgroups a fatter
    // instantiation takes the longer slices.
#define RLVI_TB(E_, B_)                                                                           \
    do {                                                                                          \
        if (launched) break;                                                                      \
        auto kern = estep_trajb_kernel<E_, B_>;                                                   \
        int G = coop_cap(kern, B_);                                                               \
        if (G > TB_G) G = TB_G;                                                                   \
        if (G >= TJ_MAXK && (N + G - 1) / G <= (int64_t)(E_) * (B_)) {                            \
            if (!dry_run)                                                                         \
                *rc = launch(kern, dim3((unsigned)G), dim3(B_), 0, st, res, wts, N, tol, maxiter,           \
                             out_iters, trace, ws, mstep_out, mstep_scale, dbg, G, Nall, pt, verify); \
            else                                                                                  \
                *rc = 0;                                                                          \
            launched = 1;                                                                         \
        }                                                                                         \
    } while (0)
    // Sharded over several GPUs: ONLY the 256-thread instantiations.  The first round's record width
    // (7 granules with the third- and fourth-order sums, rlvi_trajb.h) and the recurrence variant hang on
    // the workgroup size, and the ranks' shards -- hence their geometries -- may differ: with one
    // workgroup size everywhere every rank pushes and polls the same records and runs the same chain.
    const bool only256 = sharded || blk == 256;
    const bool only512 = !sharded && blk == 512;
    if (!only512) {
        RLVI_TB(1, 256); RLVI_TB(2, 256); RLVI_TB(3, 256); RLVI_TB(4, 256); RLVI_TB(6, 256); RLVI_TB(8, 256);
        RLVI_TB(10, 256); RLVI_TB(12, 256); RLVI_TB(16, 256);
    }
    if (!only256) {
        RLVI_TB(2, 512); RLVI_TB(4, 512); RLVI_TB(6, 512); RLVI_TB(8, 512); RLVI_TB(12, 512); RLVI_TB(16, 512);
    }
    if (!only512) { RLVI_TB(24, 256); RLVI_TB(32, 256); }
#undef RLVI_TB
    return launched;
}

}  // namespace rlvi
