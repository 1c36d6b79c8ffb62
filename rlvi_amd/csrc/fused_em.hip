// In-batch E+M in ONE launch (online-learning/main.py:296-299 order: losses of the batch -> E-step on
// them -> the weighted gradient with the NEW pi; train_rlvi.py:14-38 for the E-step, :85-96 for the
// loss and its gradient).  The whole logit block stays on the chip between the two passes:
//
//   1  every wave streams two 16-row tiles of the block into its own slice of LDS (nontemporal
//      16-B/lane reads, as mstep.hip's wave-tile form), takes max / sum exp / NLL / top-1 per row and
//      leaves exp(z - max) in place.  A workgroup of 8 waves holds 256 rows = 102 KiB of LDS at
//      C = 100: one workgroup per CU, 256 workgroups hold 65 536 x 100 fp32 (26 MB of the chip's
//      40 MB of LDS);
//   2  the 256 NLLs of a workgroup ARE its slice of the E-step: waves 0..3 run the cooperating
//      trajectory solve (rlvi_trajb.h) with one sample per thread, handed over through LDS -- no
//      second read of anything from HBM, no kernel boundary;
//   3  every wave turns its rows' NLLs into pi with the solve's (r, min), scales the resident tile
//      (flat 16-B chunks: a chunk never straddles rows because 4 | C) and streams the gradient out
//      with nontemporal stores; loss rows (l - min) and pi go out beside it; workgroup 0 gathers
//      the 256 per-workgroup records into `out`.
//
// Same arithmetic, operation for operation, as the three-launch composition in aux.hip
// (rlvi_mstep_fwd_bwd_f32 -> rlvi_estep_deep_f32 -> rlvi_mstep_fwd_bwd_f32): pi, the loss rows and
// the gradient are bit-identical to it where the E-step slices coincide (tests/test_gpu_parity.py).
// Dispatch (try_launch_fused_em, in this order): C <= 16: a row per thread, rows in REGISTERS (64 <= B <= 65 536);
// 4 | C, 16 < C <= 128, 64 <= B <= 16 384: four lanes per row, rows in registers; then this LDS-resident kernel:
// fp32, dense rows, 4 | C, 32 <= C <= 128 (four lanes per row there too), 16 | B, 64 <= ceil(B/256) <= the
// co-resident workgroups of the device -- i.e. 16 385 <= B <= 65 536, and 16 129 <= B <= 16 384 when the register
// form is refused for lack of co-resident workgroups.  Everything else (bf16, rows that are no multiple of four
// columns, strided rows, more than 65 536 rows) takes the composition.
#include "rlvi_trajb.h"

namespace rlvi {

typedef unsigned int fe_vu4 __attribute__((ext_vector_type(4)));

constexpr int FE_WAVES = 8;                      // waves per workgroup
constexpr int FE_THREADS = FE_WAVES * WAVE;
constexpr int FE_TPW = 2;                        // tiles per wave
constexpr int FE_R = 16;                         // rows per tile: four lanes per row
constexpr int FE_ROWS = FE_WAVES * FE_TPW * FE_R;     // rows per workgroup = E-step slice
constexpr int FE_EB = 256;                       // E-step threads (one sample each)
static_assert(FE_ROWS == FE_EB, "a workgroup's rows are its E-step slice, one sample per thread");

struct __attribute__((aligned(16))) FeRow { float inv_s, pw; int y; int pad; };

// softmax entry -> gradient entry: e * inv_s, and for the label column -pi * inv_scale on top of the
// ROUNDED product (mstep.hip's wave-tile form does that as a read-modify-write of the stored entry)
__device__ __forceinline__ float fe_grad(float e, float inv_s, bool label, float pw, float inv_scale) {
    float p;
    {
#pragma clang fp contract(off)
        p = e * inv_s;
    }
    return label ? fmaf(-inv_scale, pw, p) : p;
}

// The batch's four scalars: one self-tagged record per workgroup (sc1 stores, one granule per lane),
// workgroup 0 gathers the G records and adds them up in a fixed order.  (No release fence anywhere: a fence
// would wait for this CU's share of the gradient stores to drain.)  All NW waves of the workgroup call;
// acc / hits: this thread's share of sum pi*l and of the top-1 hits (a row counted by one thread).
template <int NW>
__device__ __forceinline__ void fe_batch_scalars(float acc, float hits, const TbSolved &sol, int64_t B,
                                                 float inv_scale, float *__restrict__ out, void *ws, int b, int G) {
    static_assert(NW * WAVE >= TB_G, "workgroup 0 reads one record per thread");
    __shared__ double red[2 * NW];
    __shared__ double fin[NW][PART_STRIDE];
    __shared__ int fin_dead;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);
    const double a = wave_sum((double)acc);
    const double h = wave_sum((double)hits);
    if (lane == 0) { red[2 * wave] = a; red[2 * wave + 1] = h; }
    __syncthreads();
    const double inv_rows100 = 100.0 / (double)B;
    double ta = 0.0, th = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { ta += red[2 * w]; th += red[2 * w + 1]; }
    const double rec[PART_STRIDE] = {ta * (double)inv_scale, th * inv_rows100, ta, th};
    if (out == nullptr) {
        // accumulate for rlvi_epoch_end_f32, as rlvi_mstep_fwd_bwd_f32 without `out` does
        if (tid == 0) {
            double *p = reinterpret_cast<double *>(static_cast<char *>(ws) + WS_PART_OFF) + (size_t)PART_STRIDE * b;
#pragma unroll
            for (int c = 0; c < PART_STRIDE; ++c) p[c] += rec[c];
        }
        return;
    }
    if (sol.dead) return;                 // (RLVI_ST_TIMEOUT is up: the host raises, `out` stays)
    gu64 *frec = (gu64 *)(reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_FEREC_OFF));
    const uint32_t ftag = sol.tag_free;
    if (tid < 8) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(
            (tid >> 1) == 0 ? rec[0] : (tid >> 1) == 1 ? rec[1] : (tid >> 1) == 2 ? rec[2] : rec[3]);
        const uint32_t half = (tid & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
        __hip_atomic_store(frec + (size_t)b * 8 + tid, ((unsigned long long)ftag << 32) | half,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b != 0) return;
    if (tid == 0) fin_dead = 0;
    __syncthreads();
    double t[PART_STRIDE] = {0.0, 0.0, 0.0, 0.0};
    if (wave < (G + WAVE - 1) / WAVE) {
        const bool mine = tid < G;
        const unsigned long long pa = (unsigned long long)(uintptr_t)(frec + (size_t)(mine ? tid : 0) * 8);
        const unsigned long long t0 = wall_clock64();
        const unsigned long long spin_ticks = spin_bound(hdr);
        fe_vu4 r0, r1, r2, r3;
        bool timeout = false;
        for (unsigned spin = 0;; ++spin) {
            asm volatile(
                "global_load_dwordx4 %0, %4, off sc1\n\t"
                "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                : "v"(pa)
                : "memory");
            const bool ok = !mine || (r0.y == ftag && r0.w == ftag && r1.y == ftag && r1.w == ftag &&
                                      r2.y == ftag && r2.w == ftag && r3.y == ftag && r3.w == ftag);
            if (__all(ok)) break;
            if ((spin & 63u) == 63u && wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
        }
        if (timeout) {
            if (lane == 0) { atomicOr(&hdr->status, RLVI_ST_TIMEOUT); fin_dead = 1; }
        } else if (mine) {
            t[0] = __longlong_as_double((long long)(((unsigned long long)r0.z << 32) | r0.x));
            t[1] = __longlong_as_double((long long)(((unsigned long long)r1.z << 32) | r1.x));
            t[2] = __longlong_as_double((long long)(((unsigned long long)r2.z << 32) | r2.x));
            t[3] = __longlong_as_double((long long)(((unsigned long long)r3.z << 32) | r3.x));
        }
    }
#pragma unroll
    for (int c = 0; c < PART_STRIDE; ++c) {
        t[c] = wave_sum(t[c]);
        if (lane == 0) fin[wave][c] = t[c];
    }
    __syncthreads();
    if (tid == 0 && fin_dead == 0) {
        double r[PART_STRIDE] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int c = 0; c < PART_STRIDE; ++c) r[c] += fin[w][c];
        out[0] = (float)r[0]; out[1] = (float)r[1]; out[2] = (float)r[2]; out[3] = (float)r[3];
    }
}

template <int KMAX, bool EXACT>
__global__ __launch_bounds__(FE_THREADS) void fused_em_kernel(
    const float *__restrict__ logits, const int64_t *__restrict__ labels, float *__restrict__ loss_rows,
    float *__restrict__ pi, int64_t B, int C, float inv_scale, float tol, int K,
    float *__restrict__ grad, float *__restrict__ out, int32_t *__restrict__ out_iters, void *ws,
    unsigned long long *__restrict__ dbg, int G, int verify) {
    constexpr int V = 4, LG = 4;                                  // floats per lane vector, lanes per row
    constexpr int NI = KMAX;                                      // 1-KiB pieces per tile (max)
    constexpr int WTILE = NI * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [FE_WAVES][FE_TPW][WTILE]
    __shared__ TbShared<FE_EB / WAVE, tb_stage(1, FE_EB)> sh;
    __shared__ float nll[FE_ROWS];
    __shared__ FeRow rowinfo[FE_WAVES][FE_TPW][FE_R];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int g = lane & (LG - 1);
    const int sub = lane / LG;
    const int b = (int)blockIdx.x;
    const int nv = C / V;                                         // vectors per row
    const int nchunk = (FE_R * C * 4) >> 4;                       // 16-byte chunks of a tile
    char *wbase = smem + (size_t)wave * FE_TPW * WTILE;
    WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);

    // RLVI_TJ_DEBUG: wall-clock stamps (100 MHz) of workgroup 0's phases, behind the solve's own
#define FE_STAMP(k) do { if ((RLVI_STAMPS && dbg != nullptr) && b == 0 && tid == 0) dbg[970 + (k)] = wall_clock64(); } while (0)
    FE_STAMP(0);
    const int64_t wg_row0 = (int64_t)b * FE_ROWS;
    const int64_t wrow0 = wg_row0 + (int64_t)wave * FE_TPW * FE_R;

    // ---- per-lane geometry (loop-invariant)
    unsigned dma_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int c = i * WAVE + lane;
        if (!(EXACT && i < NI - 1)) c = c < nchunk ? c : nchunk - 1;   // past the tile: its last chunk again
        dma_off[i] = (unsigned)c * 16u;
    }
    int slot_off[KMAX];
    bool live[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int vec = k * LG + g;
        live[k] = (EXACT && k < KMAX - 1) || vec < nv;
        slot_off[k] = (sub * C + (live[k] ? vec : nv - 1) * V) * 4;
    }
    const int row_off = sub * C * 4;

    // ---- 1a  loads: the caller's pi (D_0 of the E-step) and the warm-start state first (loads return
    // in order: behind the tiles they would hold tile 0 back until tile 1 has landed), labels, both tiles
    const bool active = true;                     // two sampling groups of 256 threads (rlvi_trajb.h)
    const int etid = tid & (FE_EB - 1);
    const int64_t erow = wg_row0 + etid;
    float q0[1];
    q0[0] = erow < B ? pi[erow] : 0.0f;
    const TbWarm wm = tb_warm(ws, B, K);
    bool tile_ok[FE_TPW];
    int64_t y64[FE_TPW];
    fe_vu4 stg[FE_TPW][NI];
#pragma unroll
    for (int j = 0; j < FE_TPW; ++j) {
        const int64_t r0 = wrow0 + (int64_t)j * FE_R;
        tile_ok[j] = r0 < B;                                      // 16 | B: a tile is whole or absent
        y64[j] = 0;
        if (tile_ok[j]) {
            y64[j] = labels[r0 + sub];
            const char *src = reinterpret_cast<const char *>(logits + r0 * C);
#pragma unroll
            for (int i = 0; i < NI; ++i)
                stg[j][i] = __builtin_nontemporal_load(reinterpret_cast<const fe_vu4 *>(src + dma_off[i]));
        }
    }
    FE_STAMP(1);   // loads issued
    // (phase 3's chunk -> row / column map: integer divisions, behind the loads)
    int rowc[NI], colc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (int)(dma_off[i] >> 4);
        rowc[i] = c / nv;
        colc[i] = (c - rowc[i] * nv) * V;
    }

    // ---- 1b  per row: max, sum exp, NLL, top-1; exp(z - max) stays in the tile
    float li[FE_TPW], ssum[FE_TPW];
    int yl[FE_TPW];
    bool okrow[FE_TPW];
    float hits = 0.0f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < FE_TPW; ++j) {
        li[j] = __builtin_inff();
        ssum[j] = 1.0f;
        yl[j] = 0;
        okrow[j] = false;
        if (!tile_ok[j]) {
            if (g == 0) nll[(wave * FE_TPW + j) * FE_R + sub] = __builtin_inff();
            continue;
        }
        char *wtile = wbase + j * WTILE;
        {
            // (tile j only: loads return in order, so tile 0 is worked on while tile 1 is in flight)
            fe_vu4 *t16 = reinterpret_cast<fe_vu4 *>(wtile);
#pragma unroll
            for (int i = 0; i < NI; ++i) t16[i * WAVE + lane] = stg[j][i];
        }
        __builtin_amdgcn_wave_barrier();
        okrow[j] = true;
        if (y64[j] < 0 || y64[j] >= C) { y64[j] = 0; okrow[j] = false; }
        bad = bad || !okrow[j];
        const int y = (int)y64[j];
        yl[j] = y;
        const float zy = *reinterpret_cast<const float *>(wtile + row_off + y * 4);
        float v[KMAX][V];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            *reinterpret_cast<float4 *>(v[k]) = *reinterpret_cast<const float4 *>(wtile + slot_off[k]);
        float m = v[0][0];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) m = __builtin_fmaxf(m, v[k][e]);
        m = group_allreduce<LG>(m, [](float a, float c) { return __builtin_fmaxf(a, c); });
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float ex = mexp(v[k][e] - m);
                v[k][e] = ex;
                s += live[k] ? ex : 0.0f;
            }
        s = group_sum<LG>(s);
        float l = __builtin_amdgcn_logf(s) * 0.69314718055994530942f - (zy - m);
        bool hit = zy == m;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit && s >= 2.0f) != 0, 0)) {
            // two exact maxima: the label counts only if it is the FIRST column at the maximum
            // (torch.max order, deep-learning/utils.py:58); the tile still holds the logits here
            int earlier = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const float4 z = *reinterpret_cast<const float4 *>(wtile + slot_off[k]);
                const int col = (k * LG + g) * V;
                const float zz[V] = {z.x, z.y, z.z, z.w};
#pragma unroll
                for (int e = 0; e < V; ++e) earlier += (live[k] && zz[e] == m && col + e < y) ? 1 : 0;
            }
            earlier = group_allreduce<LG>(earlier, FAdd());
            hit = hit && earlier == 0;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            *reinterpret_cast<float4 *>(wtile + slot_off[k]) = *reinterpret_cast<const float4 *>(v[k]);
        // a rejected row keeps the loss it had (the composition's forward pass skips it too)
        if (!okrow[j]) l = loss_rows[wrow0 + j * FE_R + sub];
        li[j] = l;
        ssum[j] = s;
        hits += (hit && okrow[j]) ? 1.0f : 0.0f;
        if (g == 0) nll[(wave * FE_TPW + j) * FE_R + sub] = l;
    }
    if (bad) atomicOr(&hdr->status, RLVI_ST_RANGE);
    FE_STAMP(2);   // rows done
    __syncthreads();
    FE_STAMP(3);

    // ---- 2  E-step on the 256 NLLs of this workgroup (all waves take per-node sums, waves 0..3 the rest)
    float l1[1], ev1[1];
    l1[0] = erow < B ? nll[etid] : __builtin_inff();
    const TbSolved sol = trajb_solve<1, FE_EB, FE_THREADS / FE_EB>(sh, wm, l1, q0, ev1, active, b, G, B, tol, K, out_iters,
                                               nullptr, ws, dbg, nullptr, verify != 0);
    const float pmax = tb_pmax(sol);
    FE_STAMP(4);   // solved

    // ---- 3  pi, loss rows, the weighted gradient from the resident tile
    float acc = 0.0f;
    if (!sol.dead) {
#pragma unroll
        for (int j = 0; j < FE_TPW; ++j) {
            if (!tile_ok[j]) continue;
            const int64_t row = wrow0 + j * FE_R + sub;
            const float lv = li[j] - sol.gmin;                      // residuals.sub_(min) (:27)
            const float w = tb_weight(sol, pmax, expf(-lv));        // (:28, :30, :38)
            const float pw = okrow[j] ? w : 0.0f;                   // a rejected row: zero gradient
            const float gs = pw * inv_scale;
            if (g == 0) {
                loss_rows[row] = lv;
                pi[row] = w;
                FeRow ri;
                ri.inv_s = gs * __builtin_amdgcn_rcpf(ssum[j]);
                ri.pw = pw;
                ri.y = yl[j];
                ri.pad = 0;
                rowinfo[wave][j][sub] = ri;
            }
            acc += okrow[j] ? li[j] * pw : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < FE_TPW; ++j) {
            if (!tile_ok[j]) continue;
            const fe_vu4 *t16 = reinterpret_cast<const fe_vu4 *>(wbase + j * WTILE);
            char *gdst = reinterpret_cast<char *>(grad + (wrow0 + (int64_t)j * FE_R) * C);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const fe_vu4 raw = t16[dma_off[i] >> 4];
                const FeRow ri = rowinfo[wave][j][rowc[i]];
                const int d = ri.y - colc[i];
                fe_vu4 o;
                o.x = __float_as_uint(fe_grad(__uint_as_float(raw.x), ri.inv_s, d == 0, ri.pw, inv_scale));
                o.y = __float_as_uint(fe_grad(__uint_as_float(raw.y), ri.inv_s, d == 1, ri.pw, inv_scale));
                o.z = __float_as_uint(fe_grad(__uint_as_float(raw.z), ri.inv_s, d == 2, ri.pw, inv_scale));
                o.w = __float_as_uint(fe_grad(__uint_as_float(raw.w), ri.inv_s, d == 3, ri.pw, inv_scale));
                if ((EXACT && i < NI - 1) || i * WAVE + lane < nchunk)
                    __builtin_nontemporal_store(o, reinterpret_cast<fe_vu4 *>(gdst + dma_off[i]));
            }
        }
    }

    FE_STAMP(5);   // gradient stores issued
    fe_batch_scalars<FE_WAVES>(g == 0 ? acc : 0.0f, g == 0 ? hits : 0.0f, sol, B, inv_scale, out, ws, b, G);
    FE_STAMP(7);   // out written
}

// ---------------------------------------------------------------------------------------
// Short rows (C <= 16: the ten classes of MNIST / CIFAR-10, SURVEY cfg3 / cfg4): the same one launch with a ROW
// PER THREAD.  A row is at most 16 registers, so nothing goes through LDS: a thread takes its row's max /
// sum exp / NLL / top-1, the row's NLL IS the thread's sample of the trajectory solve (the grid and slices of
// the stand-alone E-step: up to 256 workgroups of 256 threads, ceil(B / G) rows each), and the gradient comes
// out of the registers the logits went into.  Column order inside a thread, so the NLL can differ in its last
// bit from the M-step kernel's two-lanes-per-row sum; the composition stays the reference in the tests (pi,
// loss rows and gradient to 1e-5, iteration count equal).  64 <= B <= 65 536, fp32, dense rows.
// ---------------------------------------------------------------------------------------
constexpr int FR_CMAX = 16;
constexpr int FR_THREADS = 256;

__global__ __launch_bounds__(FR_THREADS) void fused_em_rows_kernel(
    const float *__restrict__ logits, const int64_t *__restrict__ labels, float *__restrict__ loss_rows,
    float *__restrict__ pi, int64_t B, int C, float inv_scale, float tol, int K,
    float *__restrict__ grad, float *__restrict__ out, int32_t *__restrict__ out_iters, void *ws,
    unsigned long long *__restrict__ dbg, int G, int verify) {
    __shared__ TbShared<FR_THREADS / WAVE, tb_stage(1, FR_THREADS)> sh;
    const int tid = threadIdx.x;
    const int b = (int)blockIdx.x;
    WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);
    const int64_t L = (B + G - 1) / G;
    const int64_t lo = (int64_t)b * L < B ? (int64_t)b * L : B;
    const int64_t hi = lo + L < B ? lo + L : B;
    const int64_t row = lo + tid;
    const bool have = row < hi;
    const TbWarm wm = tb_warm(ws, B, K);

    // ---- 1  the row: max, sum exp, NLL, top-1 (train_rlvi.py:89, utils.py:58); exp(z - max) stays in z[]
    float z[FR_CMAX];
    float q0[1], l1[1], ev1[1];
    q0[0] = have ? pi[row] : 0.0f;                                  // the caller's pi: D_0 of the E-step
    int64_t y64 = have ? labels[row] : 0;
#pragma unroll
    for (int c = 0; c < FR_CMAX; ++c) z[c] = (have && c < C) ? logits[row * C + c] : -__builtin_inff();
    bool okrow = have;
    if (have && (y64 < 0 || y64 >= C)) { y64 = 0; okrow = false; }
    if (have && !okrow) atomicOr(&hdr->status, RLVI_ST_RANGE);
    const int y = (int)y64;
    float m = z[0], zy = z[0];
    int earlier = 0;
#pragma unroll
    for (int c = 1; c < FR_CMAX; ++c) m = __builtin_fmaxf(m, z[c]);
#pragma unroll
    for (int c = 0; c < FR_CMAX; ++c) {
        zy = c == y ? z[c] : zy;
        earlier += (c < y && z[c] == m) ? 1 : 0;                    // (torch.max: the FIRST column at the maximum)
    }
    float ssum = 0.0f;
#pragma unroll
    for (int c = 0; c < FR_CMAX; ++c) {
        z[c] = mexp(z[c] - m);                                      // (columns past C: exp(-inf) = 0)
        ssum += z[c];
    }
    float l = __builtin_amdgcn_logf(ssum) * 0.69314718055994530942f - (zy - m);
    const bool hit = okrow && zy == m && earlier == 0;
    // a rejected row keeps the loss it had (the composition's forward pass skips it too)
    if (have && !okrow) l = loss_rows[row];
    l1[0] = have ? l : __builtin_inff();

    // ---- 2  E-step on the workgroup's rows: one sample per thread, as estep_trajb_kernel<1, 256>
    const TbSolved sol = trajb_solve<1, FR_THREADS>(sh, wm, l1, q0, ev1, true, b, G, B, tol, K, out_iters, nullptr,
                                                    ws, dbg, nullptr, verify != 0);
    const float pmax = tb_pmax(sol);

    // ---- 3  pi, the loss row, the weighted gradient out of the registers
    float acc = 0.0f;
    if (!sol.dead && have) {
        const float w = tb_weight(sol, pmax, ev1[0]);               // (:28, :30, :38)
        const float pw = okrow ? w : 0.0f;                          // a rejected row: zero gradient
        const float inv_s = (pw * inv_scale) * __builtin_amdgcn_rcpf(ssum);
        loss_rows[row] = l - sol.gmin;                              // residuals.sub_(min) (:27)
        pi[row] = w;
#pragma unroll
        for (int c = 0; c < FR_CMAX; ++c)
            if (c < C) grad[row * C + c] = fe_grad(z[c], inv_s, c == y, pw, inv_scale);
        acc = okrow ? l * pw : 0.0f;
    }
    fe_batch_scalars<FR_THREADS / WAVE>(acc, hit ? 1.0f : 0.0f, sol, B, inv_scale, out, ws, b, G);
}

// ---------------------------------------------------------------------------------------
// Wide rows, small batches (4 | C, 16 < C <= 128, 64 <= B <= 16 384: too few rows to give every CU the 256 the
// LDS-resident kernel above wants, more columns than a thread's registers hold): four lanes per row as in the
// M-step kernel, the row's vectors k * 4 + g in lane g's registers from the first read to the gradient's store,
// up to 64 rows per workgroup of 256 threads.  Lane 0 of a row's group carries the row's NLL into the trajectory solve
// as its sample (the other lanes hold pads, which drop out of every sum).
// ---------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(FR_THREADS) void fused_em_rows4_kernel(
    const float *__restrict__ logits, const int64_t *__restrict__ labels, float *__restrict__ loss_rows,
    float *__restrict__ pi, int64_t B, int C, float inv_scale, float tol, int K,
    float *__restrict__ grad, float *__restrict__ out, int32_t *__restrict__ out_iters, void *ws,
    unsigned long long *__restrict__ dbg, int G, int verify) {
    constexpr int V = 4, LG = 4;
    __shared__ TbShared<FR_THREADS / WAVE, tb_stage(1, FR_THREADS)> sh;
    const int tid = threadIdx.x;
    const int g = tid & (LG - 1);
    const int b = (int)blockIdx.x;
    WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);
    const int64_t L = (B + G - 1) / G;                               // rows of a workgroup (<= ROWS: the launcher)
    const int64_t row = (int64_t)b * L + (tid >> 2);
    const bool have = (tid >> 2) < L && row < B;
    const int nv = C / V;
    const TbWarm wm = tb_warm(ws, B, K);

    // ---- 1  the row: max, sum exp, NLL, top-1; exp(z - max) stays in v[][]
    float q0[1], l1[1], ev1[1];
    q0[0] = (have && g == 0) ? pi[row] : 0.0f;
    int64_t y64 = have ? labels[row] : 0;
    float v[KMAX][V];
    bool live[KMAX];
    const float *zrow = logits + (have ? row : 0) * C;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int vec = k * LG + g;
        live[k] = vec < nv;
        const float4 t = *reinterpret_cast<const float4 *>(zrow + (live[k] ? vec : 0) * V);
        v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
    bool okrow = have;
    if (have && (y64 < 0 || y64 >= C)) { y64 = 0; okrow = false; }
    if (have && !okrow && g == 0) atomicOr(&hdr->status, RLVI_ST_RANGE);
    const int y = (int)y64;
    const float zy = zrow[y];
    float m = v[0][0];                                               // (lane g's vector 0 exists: nv >= 4)
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) m = live[k] ? __builtin_fmaxf(m, v[k][e]) : m;
    m = group_allreduce<LG>(m, [](float a, float c) { return __builtin_fmaxf(a, c); });
    int earlier = 0;
    float ssum = 0.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int col = (k * LG + g) * V + e;
            earlier += (live[k] && v[k][e] == m && col < y) ? 1 : 0;   // (torch.max: the FIRST column at the maximum)
            const float ex = mexp(v[k][e] - m);
            v[k][e] = ex;
            ssum += live[k] ? ex : 0.0f;
        }
    ssum = group_sum<LG>(ssum);
    earlier = group_allreduce<LG>(earlier, FAdd());
    float l = __builtin_amdgcn_logf(ssum) * 0.69314718055994530942f - (zy - m);
    const bool hit = okrow && zy == m && earlier == 0;
    // a rejected row keeps the loss it had (the composition's forward pass skips it too)
    if (have && !okrow) l = loss_rows[row];
    l1[0] = (have && g == 0) ? l : __builtin_inff();

    // ---- 2  E-step: the rows of this workgroup are its slice, one sample in every fourth thread
    const TbSolved sol = trajb_solve<1, FR_THREADS>(sh, wm, l1, q0, ev1, true, b, G, B, tol, K, out_iters, nullptr,
                                                    ws, dbg, nullptr, verify != 0);
    const float pmax = tb_pmax(sol);

    // ---- 3  pi, the loss row, the weighted gradient out of the registers
    float acc = 0.0f;
    if (!sol.dead && have) {
        const float lv = l - sol.gmin;                               // residuals.sub_(min) (:27)
        const float w = tb_weight(sol, pmax, expf(-lv));             // (:28, :30, :38)
        const float pw = okrow ? w : 0.0f;                           // a rejected row: zero gradient
        const float inv_s = (pw * inv_scale) * __builtin_amdgcn_rcpf(ssum);
        if (g == 0) { loss_rows[row] = lv; pi[row] = w; acc = okrow ? l * pw : 0.0f; }
        float *grow = grad + row * C;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (!live[k]) continue;
            const int col = (k * LG + g) * V;
            float4 o;
            o.x = fe_grad(v[k][0], inv_s, col == y, pw, inv_scale);
            o.y = fe_grad(v[k][1], inv_s, col + 1 == y, pw, inv_scale);
            o.z = fe_grad(v[k][2], inv_s, col + 2 == y, pw, inv_scale);
            o.w = fe_grad(v[k][3], inv_s, col + 3 == y, pw, inv_scale);
            *reinterpret_cast<float4 *>(grow + col) = o;
        }
    }
    fe_batch_scalars<FR_THREADS / WAVE>(acc, (hit && g == 0) ? 1.0f : 0.0f, sol, B, inv_scale, out, ws, b, G);
}

// Eligibility + launch.  Returns 1 if launched (rc in *rc), 0 if the composition has to take it.
int try_launch_fused_em(const float *logits, int64_t ld, const int64_t *labels, float *loss_rows, float *pi,
                        int64_t B, int64_t C, float inv_scale, float tol, int maxiter, float *grad,
                        int64_t ldg, float *out, int32_t *out_iters, void *ws, hipStream_t st, int *rc) {
    if (tune_get("RLVI_FUSED_EM", 1) == 0) return 0;
    if (grad != nullptr && ld == C && ldg == C && C <= FR_CMAX && B >= 64 && maxiter >= 1 && maxiter <= TJ_MAXK) {
        // short rows: a row per thread, the stand-alone E-step's grid (its admission rule too: every
        // exchanging workgroup co-resident, node k reduced by workgroup k)
        auto kern = fused_em_rows_kernel;
        int G = coop_cap(kern, FR_THREADS);
        if (G > TB_G) G = TB_G;
        if (G >= TJ_MAXK && (B + G - 1) / G <= FR_THREADS) {
            const int debug = tune_get("RLVI_TJ_DEBUG", 0);
            unsigned long long *dbg = debug ? reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF) : nullptr;
            *rc = launch(kern, dim3((unsigned)G), dim3(FR_THREADS), 0, st, logits, labels, loss_rows, pi, B, (int)C,
                         inv_scale, tol, maxiter, grad, out, out_iters, ws, dbg, G, tune_get("RLVI_TJ_VERIFY", 0));
            return 1;
        }
        return 0;
    }
    if (grad != nullptr && ld == C && ldg == C && !(C & 3) && C > FR_CMAX && C <= 128 && B >= 64 &&
        B <= (int64_t)TB_G * (FR_THREADS / 4) && maxiter >= 1 && maxiter <= TJ_MAXK &&
        !(((uintptr_t)logits & 15) || ((uintptr_t)grad & 15))) {
        // wide rows, too few of them for the LDS-resident kernel below: four lanes per row, up to 64 rows per
        // workgroup on as many workgroups as are co-resident (at most 256)
        const int debug = tune_get("RLVI_TJ_DEBUG", 0);
        unsigned long long *dbg = debug ? reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF) : nullptr;
        const int verify = tune_get("RLVI_TJ_VERIFY", 0);
        auto go = [&](auto kern) {
            int G = coop_cap(kern, FR_THREADS);
            if (G > TB_G) G = TB_G;
            if (G < TJ_MAXK || (B + G - 1) / G > FR_THREADS / 4) return 0;   // node k is reduced by workgroup k; 64 rows each
            *rc = launch(kern, dim3((unsigned)G), dim3(FR_THREADS), 0, st, logits, labels, loss_rows, pi, B, (int)C,
                         inv_scale, tol, maxiter, grad, out, out_iters, ws, dbg, G, verify);
            return 1;
        };
        if (C <= 64 ? go(fused_em_rows4_kernel<4>) : go(fused_em_rows4_kernel<8>)) return 1;
        // (refused -- fewer than 64 co-resident workgroups of that kernel, e.g. under RLVI_DEVICE_SHARERS: the
        //  LDS-resident kernel below may still take the shape before the three-launch composition does)
    }
    if (grad == nullptr || ld != C || ldg != C || (C & 3) || C < 32 || C > 128 || (B & 15)) return 0;
    if (((uintptr_t)logits & 15) || ((uintptr_t)grad & 15)) return 0;
    if (maxiter < 1 || maxiter > TJ_MAXK) return 0;
    const int64_t G64 = (B + FE_ROWS - 1) / FE_ROWS;
    if (G64 < TJ_MAXK || G64 > TB_G) return 0;      // node k of the trajectory is reduced by workgroup k
    const int G = (int)G64;
    const int nv = (int)C / 4, k = (nv + 3) / 4;
    const int debug = tune_get("RLVI_TJ_DEBUG", 0);
    unsigned long long *dbg = debug ? reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + WS_SCRATCH_OFF) : nullptr;
    const int verify = tune_get("RLVI_TJ_VERIFY", 0);
    int launched = 0;
#define RLVI_FE(K_, X_)                                                                              \
    do {                                                                                             \
        auto kern = fused_em_kernel<K_, X_>;                                                         \
        const size_t lds = (size_t)FE_WAVES * FE_TPW * (K_) * 1024;                                  \
        /* > 64 KiB of dynamic LDS has to be asked for, once per device of this process */          \
        static int attr_dev = -1;                                                                    \
        int cur_dev = 0;                                                                             \
        if (hipGetDevice(&cur_dev) != hipSuccess) break;                                             \
        if (attr_dev != cur_dev) {                                                                   \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                break;                                                                               \
            attr_dev = cur_dev;                                                                      \
        }                                                                                            \
        if (coop_cap(kern, FE_THREADS, lds) < G) break;     /* all G workgroups must be resident */  \
        *rc = launch(kern, dim3((unsigned)G), dim3(FE_THREADS), lds, st, logits, labels, loss_rows, pi, B, \
                     (int)C, inv_scale, tol, maxiter, grad, out, out_iters, ws, dbg, G, verify);     \
        launched = 1;                                                                                \
    } while (0)
    if (k <= 4) RLVI_FE(4, false);
    else if (k == 7) RLVI_FE(7, true);
    else RLVI_FE(8, false);
#undef RLVI_FE
    return launched;
}

}  // namespace rlvi
