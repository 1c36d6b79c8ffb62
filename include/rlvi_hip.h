/*
 * rlvi_hip.h -- C ABI of librlvi_gfx950.so, the MI355X (gfx950 / CDNA4) implementation
 * of the RLVI E-step / M-step hot path.
 *
 * The reference (akarakulev/rlvi) is pure Python; its "FFI" for this path is the set of
 * torch / numpy calls inside the plug-in function
 *     methods.train_rlvi(train_loader, model, optimizer, residuals, weights, overfit, threshold)
 * (deep-learning/methods/train_rlvi.py:52-106, selected by --method=rlvi, main.py:26,277-280)
 * and its two helpers update_sample_weights (:14-38) / false_negative_criterion (:41-49),
 * plus standard-learning/rlvi.py:8-20,68-89 and online-learning/main.py:45-58,84-85.
 * Each entry point below names the reference statements it replaces.  The Python binding
 * a maintainer would add is rlvi_amd/_lib.py (ctypes); see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch "cuda" tensor) unless marked host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - functions only enqueue work: no allocation, no host synchronisation, graph-capturable;
 *   - return value: 0 = ok, < 0 = argument error (RLVI_E_*), > 0 = hipError_t of the launch;
 *   - scratch comes from a caller-owned workspace (rlvi_workspace_bytes / rlvi_workspace_init).
 *     One workspace must not be used by two streams at the same time.
 *   - device-side status: word 0 of the workspace is a sticky int32 status (0 = ok), bit flags
 *     RLVI_ST_*; out-of-range labels / indexes never touch memory, they set RLVI_ST_RANGE.
 *     The Python plug-in reads it at every epoch end and raises on any flag.
 *   - kernels whose workgroups wait for each other (E-step, threshold) size their grid from the
 *     occupancy query of the current device, so that all of them are resident at once; every wait
 *     is bounded in wall time (RLVI_SPIN_BOUND_MS, default 100) and sets RLVI_ST_TIMEOUT.
 *     The query sees THIS process only: when S processes drive the same GPU, tell each of them
 *     (rlvi_tune_set("RLVI_DEVICE_SHARERS", S), or the environment variable of that name) and every
 *     process takes 1/S of the proven capacity; rlvi_amd.dist does it for the ranks of a process
 *     group by comparing rlvi_device_pci_bus_id() over the ranks.
 */
#ifndef RLVI_HIP_H
#define RLVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLVI_ABI_VERSION 3   /* 3: calls with `out` have records of their own (workspace layout), per-workspace options,
                                 rlvi_tune_unset / _overrides, rlvi_linear_regression_f64, rlvi_sample_weight_online_f64,
                                 rlvi_workspace_region / _reset_warm, rlvi_stream_copy
                                 2: workspace layout of round 2 (peer table, sharded state), rlvi_device_pci_bus_id,
                                 rlvi_estep_sharded_check; set_peers zeroes the local inbox */

#define RLVI_E_NULL   (-1) /* required pointer is NULL                 */
#define RLVI_E_SHAPE  (-2) /* negative / inconsistent size             */
#define RLVI_E_ALIGN  (-3) /* pointer or leading dimension misaligned  */
#define RLVI_E_WS     (-4) /* workspace too small / not initialised    */
#define RLVI_E_LIMIT  (-5) /* size beyond what the kernels support     */

#define RLVI_ST_RANGE   1  /* a label or index was out of range: the row contributes nothing (zero gradient row,
                              no residual written, not counted in loss / top-1)     */
#define RLVI_ST_TIMEOUT 2  /* an inter-workgroup wait hit its bound (the workgroups of a cooperating launch were
                              not all resident, e.g. beside another process's kernels): the launch left its
                              outputs -- pi, threshold -- as they were                */
#define RLVI_ST_NOCONV  4  /* the trajectory E-step did not reach its fixed point (results invalid) */
#define RLVI_ST_SINGULAR 8 /* weighted least squares: rank-deficient design; theta is the minimum-norm
                              solution (what the reference's lstsq returns), NaN for non-finite data */

int rlvi_abi_version(void);
const char *rlvi_error_string(int code);

/* Integer tuning / debug knob (same names as the RLVI_* environment variables, which it
 * overrides); takes effect from the next launch.  Not needed for normal operation. */
int rlvi_tune_set(const char *name, int value);
/* Forget a rlvi_tune_set value (environment variable / built-in default again); 1 if there was one, else 0. */
int rlvi_tune_unset(const char *name);
/* Names of the knobs that carry a rlvi_tune_set value now, comma-separated, into buf[len]; returns their number.
 * The knobs are process-wide: a test harness asserts 0 after every test. */
int rlvi_tune_overrides(char *buf, int len);
/* Compute units of the current device as the launchers see them. */
int rlvi_device_cus(void);
/* PCI bus id ("0000:c1:00.0") of the current device into buf[len >= 16]: the identity of the physical GPU,
 * the same in every process whatever the visible-device masks made of the ordinals. */
int rlvi_device_pci_bus_id(char *buf, int len);

/* Bytes of workspace needed for vectors up to max_n samples and batches up to max_b rows. */
size_t rlvi_workspace_bytes(int64_t max_n, int64_t max_b);
/* Zero the control words.  Call once after allocating (or to clear a sticky status). */
int rlvi_workspace_init(void *ws, size_t ws_bytes, void *stream);
/* Copy the sticky status word to *status_host (synchronises the stream). */
int rlvi_workspace_status(const void *ws, int32_t *status_host, void *stream);
/* Reset the sticky status word to 0 (nothing else: warm-start state and records are kept). */
int rlvi_workspace_clear_status(void *ws, void *stream);
/* What the CALLER of this workspace tells the launchers (host side, per workspace, from the next launch on;
 * rlvi_workspace_init forgets it):
 *   "logits_from_hbm" 1: the [B, C] blocks of the M-step calls on this workspace stream from HBM (larger than the
 *       Infinity Cache, or one of many blocks touched in rotation): a one-tile-per-wave launch then holds its
 *       gradient stores for the read time of the block (mstep.hip); 0 (default): nothing is assumed;
 *   "cold_start" 1: E-step and threshold take no guess from the previous call on this workspace (every call as the
 *       reference's loop starts it, train_rlvi.py:29; the results never depend on the guesses, only the time does).
 * RLVI_E_SHAPE for an unknown name. */
int rlvi_workspace_set_option(void *ws, const char *name, int value);
/* Which form the last M-step launch on this workspace took (tests of the dispatch): 0 none yet, 1 register rows,
 * 2 wave tiles in four-wave workgroups, 3 wave tiles in 16-wave workgroups, 4 word-wise bf16 wave tiles (odd row
 * lengths), 5 long rows (more than 512 vectors per row: a wave or a workgroup per row, three passes); + 16 with a timed hold. */
int rlvi_workspace_last_mstep_form(const void *ws);
/* Forget the guesses earlier calls left for the next one (E-step trajectory and minimum, threshold key). */
int rlvi_workspace_reset_warm(void *ws, void *stream);
/* Byte offset (and size through *bytes) of a region of the workspace layout, for tools and tests: "records"
 * (accumulate-mode M-step records), "records_out" (records of calls with `out`), "warm", "scratch"; (size_t)-1
 * for an unknown name. */
size_t rlvi_workspace_region(const char *name, size_t *bytes);

/* ---------------------------------------------------------------------------------------
 * M-step over one mini-batch, forward + backward w.r.t. the logits, lagged pi.
 * Replaces train_rlvi.py:85 (top-1 of accuracy()), :89 (F.cross_entropy reduction='none'),
 * :90 (residuals[indexes] = loss), :92 (weights[indexes]), :93-94 (weighted mean) and the
 * autograd backward of those statements reached from :96.
 *
 *   logits      [B, C] row-major, leading dimension ld (elements)
 *   labels, idx [B] int64;   weights, residuals [N] fp32
 *   inv_scale   1/B for one device; 1/global_B when the batch is sharded over ranks
 *   grad_logits [B, C] (ldg) or NULL for forward only: inv_scale*pi_i*(softmax - onehot)
 *   out         [4] fp32 device: { sum_i pi_i*l_i * inv_scale, 100*hits/B, sum_i pi_i*l_i, hits }
 * bf16 variant: logits / grad_logits are bfloat16, arithmetic is fp32 on the widened values.
 * Any C up to 2^20 (RLVI_E_LIMIT beyond); rows of more than 512 vectors (C > 2048 fp32 / 4096 bf16 with aligned
 * rows, C > 512 with an odd length or pitch) take a form that reads the row three times (a wave per row, a workgroup
 * per row for batches of up to four rows per CU).
 *
 * Evaluation form: weights == NULL (then idx and residuals must be NULL too) computes the plain
 * mean CE and the top-1 percentage of the batch -- utils.evaluate (deep-learning/utils.py:48-62)
 * without a separate softmax / argmax pass.
 *
 * out == NULL selects ACCUMULATE mode: nothing is finalised; every workgroup adds its partial sums
 * {loss_b, top-1 % of the batch, sum pi*l, hits} to its own record in the workspace (no atomics),
 * so a whole epoch of mini-batches costs one launch each, and rlvi_epoch_end_f32 (or
 * rlvi_mstep_reduce_f32) reduces and clears the records.  A call WITH `out` keeps its records apart
 * (ABI 3) -- it may be interleaved with an accumulate sequence on the same workspace, e.g. an evaluation
 * batch between two training batches -- and is a second (one-workgroup) launch behind the first: +2.6 us
 * in the stream at 65 536 x 100, +1.8 us at 4096 x 10 -- a training loop that needs the batch scalars
 * only at the epoch end wants ACCUMULATE.
 */
int rlvi_mstep_fwd_bwd_f32(const float *logits, int64_t ld, const int64_t *labels,
                           const int64_t *idx, const float *weights, float *residuals,
                           int64_t N, int64_t B, int64_t C, float inv_scale,
                           float *grad_logits, int64_t ldg, float *out, void *ws, void *stream);
int rlvi_mstep_fwd_bwd_bf16(const uint16_t *logits, int64_t ld, const int64_t *labels,
                            const int64_t *idx, const float *weights, float *residuals,
                            int64_t N, int64_t B, int64_t C, float inv_scale,
                            uint16_t *grad_logits, int64_t ldg, float *out, void *ws,
                            void *stream);

/* out[4] = scale * {sum loss_b, sum top-1 %_b}, sum pi*l, hits over the accumulated batches;
 * clears the records (scale = 1/batches gives the reference's train_acc, train_rlvi.py:105). */
int rlvi_mstep_reduce_f32(float *out, double scale, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * E-step, deep-learning variant, in place on both vectors.
 * Replaces update_sample_weights(residuals, weights, tol, maxiter), train_rlvi.py:14-38:
 * min-shift of the residuals (:27), e = exp(-l) (:28), the fixed point on mean(pi) with the
 * caller's pi entering the first error only (:29-37), and the final pi /= max(pi) (:38).
 *   out_iters  device int32 (may be NULL): iterations executed
 *   trace      device fp32 [2*maxiter] (may be NULL): error then mean-pi of every iteration
 * ------------------------------------------------------------------------------------- */
int rlvi_estep_deep_f32(float *residuals, float *weights, int64_t N, float tol, int maxiter,
                        int32_t *out_iters, float *trace, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * The same E-step with the samples SHARDED over the GPUs of one node (new; the reference is
 * single-device): one process per GPU, this rank holds n_local of the n_all samples.  The
 * per-sample work stays local; only the per-node totals of the trajectory solve (<= 24 nodes x 7
 * fp32 per round) cross the GPUs, written by the kernel's reducer workgroups straight into the other
 * ranks' inboxes over xGMI and summed in rank order, so every rank reaches the same fixed point bit
 * for bit and writes pi / the min-shifted residuals of its own samples.  No collective-library call
 * and no gather of the residual vector on the path.
 *
 *   rlvi_peer_alloc / _export / _open / _close / _free   one 320-KiB inbox per rank in uncached device
 *       memory, exchanged as 64-byte IPC handles by the host program (rlvi_amd.dist.setup_peers)
 *   rlvi_workspace_set_peers(ws, rank, world, inboxes, stream)   inboxes[r] = rank r's inbox as mapped
 *       in this process; call at the same program point on every rank: it resets the round counter and
 *       the sharded warm-start state and ZEROES THE LOCAL INBOX (records of an earlier set-up must not
 *       match the new rounds' tags), so a host-side barrier over the ranks must separate it from the
 *       first sharded call (a peer must not push before this rank's inbox is clean).  After a sharded
 *       call raised RLVI_ST_TIMEOUT the ranks' round counters may differ: set the peers up again (on
 *       every rank) before the next sharded call.
 *   rlvi_estep_sharded_check(n_local, n_all, maxiter, with_out)   0 if rlvi_estep_sharded_f32 would launch
 *       for this shape on this device now, RLVI_E_LIMIT if not; nothing is launched.  The ranks compare
 *       answers before the first collective call.  The sharded solve uses the 256-thread geometry on
 *       every rank (shards may differ in length; the exchanged record layout may not).
 *   rlvi_estep_sharded_f32   a collective: every rank calls it the same number of times; returns
 *       RLVI_E_LIMIT when the shape is outside the trajectory kernel (n_local < 64, ...).
 *       out != NULL: this rank's M-step scalars of the epoch too (as rlvi_epoch_end_f32, x 1/batches).
 *       Waits on a peer are bounded by 100 x the workspace's spin bound (a peer may legitimately be
 *       late); a rank that gives up raises RLVI_ST_TIMEOUT.
 * ------------------------------------------------------------------------------------- */
#define RLVI_PEER_HANDLE_BYTES 64
size_t rlvi_peer_inbox_bytes(void);
int rlvi_peer_alloc(void **inbox);
int rlvi_peer_free(void *inbox);
int rlvi_peer_export(void *inbox, void *handle64);
int rlvi_peer_open(const void *handle64, void **inbox);
int rlvi_peer_close(void *inbox);
int rlvi_peer_can_access(int peer_device);   /* 1 / 0: the current device can map `peer_device`'s memory */
int rlvi_workspace_set_peers(void *ws, int rank, int world, void *const *inboxes, void *stream);
int rlvi_workspace_clear_peers(void *ws, void *stream);   /* before closing / freeing the inboxes */
int rlvi_estep_sharded_check(int64_t n_local, int64_t n_all, int maxiter, int with_out);
int rlvi_estep_sharded_f32(float *residuals, float *weights, int64_t n_local, int64_t n_all,
                           float tol, int maxiter, int64_t batches, float *out, int32_t *out_iters,
                           void *ws, void *stream);
/* The type-II threshold + truncation (train_rlvi.py:41-49,:102-103) on weights sharded the same way:
 * the radix descent's per-bin totals take the same route through the inboxes (exact integers, so
 * bit-exact as on one device); *thr_inout and *kept_out (over ALL ranks) are identical on every rank,
 * the truncation and mask_gt cover this rank's n_local weights.  RLVI_E_LIMIT for n_local <= 1024;
 * a weight outside [0, 1] raises RLVI_ST_NOCONV (the one-device generic form cannot see the others). */
int rlvi_threshold_sharded_check(int64_t n_local, int64_t n_all);   /* 0 / RLVI_E_LIMIT: would the call below launch? */
int rlvi_threshold_truncate_sharded_f32(float *weights, int64_t n_local, int64_t n_all, float alpha,
                                        float *thr_inout, uint8_t *mask_gt, int64_t *kept_out,
                                        void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * End of an epoch in one call, replaces train_rlvi.py:99-105: update_sample_weights over all N
 * samples; if `overfit`, *thr_inout = max(*thr_inout, criterion) and the truncation; and, if
 * out != NULL, the epoch's M-step scalars reduced from the accumulate-mode records by an extra
 * workgroup of the same launch: out[1] = mean over `batches` of the per-batch top-1 percentage
 * (= train_acc of :105), out[0] = mean batch loss.
 * ------------------------------------------------------------------------------------- */
int rlvi_epoch_end_f32(float *residuals, float *weights, int64_t N, float tol, int maxiter,
                       int overfit, float alpha, float *thr_inout, int64_t batches, float *out,
                       int32_t *out_iters, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * Type-II-error threshold, replaces false_negative_criterion(weights, alpha),
 * train_rlvi.py:41-49 (sum, sort, cumsum, count, gather) without sorting: the position is
 * found by a radix descent (8 bits per digit, most significant first) on the order-preserving key
 * with exact integer per-bin counts and sums.
 *   thr_out  device fp32: the threshold
 * rlvi_threshold_truncate_f32 additionally applies train_rlvi.py:102-103:
 *   *thr_inout = max(*thr_inout, criterion); weights[weights < *thr_inout] = 0
 * and, if mask_gt != NULL, writes mask_gt[i] = weights[i] > *thr_inout (main.py:343) and
 * kept_out = number of set mask entries (device int64, may be NULL).
 * ------------------------------------------------------------------------------------- */
int rlvi_fn_threshold_f32(const float *weights, int64_t N, float alpha, float *thr_out,
                          void *ws, void *stream);
int rlvi_threshold_truncate_f32(float *weights, int64_t N, float alpha, float *thr_inout,
                                uint8_t *mask_gt, int64_t *kept_out, void *ws, void *stream);
/* Elementwise part alone (threshold already known, on device). */
int rlvi_truncate_f32(float *weights, int64_t N, const float *thr, uint8_t *mask_gt,
                      void *stream);

/* ---------------------------------------------------------------------------------------
 * Small-loss selection (the baselines that share the "per-sample CE -> select -> mean" shape,
 * SURVEY 8(f)-4).  Replaces  ind_sorted = np.argsort(loss.cpu()); ind_update = ind_sorted[:k]
 * (train_usdnl.py:18-24, train_coteaching.py:18-30): mask_w[i] = 1.0f for the k smallest of the n
 * losses, else 0.0f; equal losses are taken in index order (a stable argsort), NaN orders last.
 * The selected rows are then weighted through rlvi_mstep_fwd_bwd_* with weights = mask_w,
 * idx = NULL and inv_scale = 1/k (usdnl) or 1/k^2 (co-teaching's mean followed by /num_remember,
 * train_coteaching.py:32-35).  k <= 0 selects nothing, k >= n everything.
 * ------------------------------------------------------------------------------------- */
int rlvi_select_smallest_f32(const float *loss, int64_t n, int64_t k, float *mask_w, void *stream);

/* ---------------------------------------------------------------------------------------
 * precision@k.  Replaces accuracy(logit, target, topk) of deep-learning/utils.py:65-79 (softmax :67, torch.topk
 * :70, eq :72, the per-k counts :76-77): hits[j] = number of rows whose label is among the ks[j]
 * largest logits of its row, j < nk <= 8.  train_rlvi keeps only precision@1 (train_rlvi.py:85), which the
 * M-step kernel delivers on the side; this is the stand-alone form.  One pass, no softmax, no sort: the label's
 * rank is the count of columns ahead of it; equal values rank in column order (torch.topk's order among equal
 * values, and logits that differ but whose fp32 softmax values coincide, are implementation details of the
 * reference: unpinned).  A label outside [0, C) is never a hit (:72).
 *   ks     HOST array of nk values, each in [1, C] (RLVI_E_SHAPE beyond: torch.topk raises there)
 *   hits   DEVICE int32 [nk], zeroed by the call; precision@k = 100 * hits / B
 * ------------------------------------------------------------------------------------- */
int rlvi_topk_hits_f32(const float *logits, int64_t ld, const int64_t *labels, int64_t B, int64_t C,
                       const int32_t *ks, int nk, int32_t *hits, void *stream);
int rlvi_topk_hits_bf16(const uint16_t *logits, int64_t ld, const int64_t *labels, int64_t B, int64_t C,
                        const int32_t *ks, int nk, int32_t *hits, void *stream);

/* ---------------------------------------------------------------------------------------
 * In-batch fused E+M (online order, online-learning/main.py:296-299 applied to a logit block):
 * per-sample NLL -> E-step on THIS batch (deep variant, pi_in only feeds the first error)
 * -> weighted loss and gradient with the NEW pi.  Composition of a1, a7, a4, a5 -- as ONE launch for fp32 dense
 * rows (fused_em.hip), tried in this order on a 256-CU device:
 *   C <= 16 (the ten classes of MNIST / CIFAR-10), 64 <= B <= 65 536: a row per thread;
 *   4 | C, 16 < C <= 128, 64 <= B <= 16 384: four lanes per row, the rows in registers;
 *   4 | C, 32 <= C <= 128, 16 | B, 16 385 <= B <= 65 536 (and 16 129 <= B <= 16 384 when the form above is refused
 *   for lack of co-resident workgroups): the logit block resident in the chip's LDS between the two passes --
 *   bit-identical to the composition where the E-step slices coincide (B > 65 280), a few ulp elsewhere.
 * The two register forms: pi, loss rows and gradient within 1e-5 of the composition, same iteration count.
 * Everything else (bf16, 4 does not divide C > 16, more than 65 536 rows, strided rows): three launches.
 *   loss_rows [B] receives the min-shifted NLL, pi [B] the new posteriors (in/out).
 * ------------------------------------------------------------------------------------- */
int rlvi_fused_em_f32(const float *logits, int64_t ld, const int64_t *labels, float *loss_rows,
                      float *pi, int64_t B, int64_t C, float inv_scale, float tol, int maxiter,
                      float *grad_logits, int64_t ldg, float *out, int32_t *out_iters,
                      void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * fp64 E-steps of the numpy paths.
 *   rlvi_update_weights_f64        standard-learning/rlvi.py:8-20   (update_weights)
 *   rlvi_update_weights_online_f64 online-learning/main.py:45-58    (update_weights_rlvi)
 * ------------------------------------------------------------------------------------- */
int rlvi_update_weights_f64(const double *losses, int64_t n, double tol, int maxiter,
                            double *out, int32_t *out_iters, void *ws, void *stream);
int rlvi_update_weights_online_f64(const double *losses, int64_t n, double tol, int maxiter,
                                   double *out, int32_t *out_iters, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * Per-sample NLL of the linear / logistic paths: the X.theta / X.w contraction on the fp64 matrix
 * cores (v_mfma_f64_16x16x4_f64, 16 rows per wave; n*d is 15-20 k elements: launch-latency-bound).
 *   rlvi_linreg_losses_f64  rlvi.py:72-74 / :81-83: r=(y-X theta)^2, sigma2=w.r/sum(w),
 *                           losses=0.5 r/sigma2;  sigma2_out device fp64
 *   rlvi_logistic_nll_f64   online-learning/main.py:295-296,:84-85: -log sigmoid(X w + b)
 * X is row-major [n, d].
 * ------------------------------------------------------------------------------------- */
/* Weighted least squares theta = argmin sum_i w_i (y_i - x_i.theta)^2, the M-step of
 * linear_regression (standard-learning/rlvi.py:70-71,:79-80, scipy lstsq on diag(sqrt(w))-scaled
 * rows there): [X | y]^T W [X | y] on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), Cholesky +
 * triangular solves in one workgroup; a rank-deficient design sets RLVI_ST_SINGULAR and gets the
 * minimum-norm solution (Jacobi eigen-decomposition of the Gram matrix), as lstsq gives.  d <= 63. */
int rlvi_wls_solve_f64(const double *X, const double *y, const double *w, int64_t n, int64_t d,
                       double *theta, void *ws, void *stream);
int rlvi_linreg_losses_f64(const double *X, const double *y, const double *theta,
                           const double *w, int64_t n, int64_t d, double *losses,
                           double *sigma2_out, void *ws, void *stream);
int rlvi_logistic_nll_f64(const double *X, const double *w, double b, int64_t n, int64_t d,
                          double *losses, void *stream);

/* ---------------------------------------------------------------------------------------
 * The whole estimator linear_regression(X, y, maxiter=100, tol=1e-3) of standard-learning/rlvi.py:68-89 as ONE
 * launch (standard.hip): weights = 1, weighted least squares (:70-71), Gaussian NLL (:72-74), then up to `maxiter`
 * times { update_weights(losses) with (estep_tol, estep_maxiter) = the reference's (1e-3, 100) (:77 -> :8-20),
 * weighted least squares (:79-80), NLL (:81-83), stop when ||theta - prev|| / ||prev|| <= tol (:85-87) } -- the
 * stop decision is taken on the device, the host waits once, for the result.  One persistent workgroup: the
 * weighted Gram matrix and X.theta on the fp64 matrix cores, an L D L^T solve on one wave.
 *   theta [d], weights [n] (the final pi), info int32[4] = { outer iterations, inner iterations of the last
 *   E-step, inner iterations in all, fallback } -- all device pointers.
 * n <= 4096 and d <= 31 (rlvi_linear_regression_check says 0 / RLVI_E_LIMIT without launching).  info[3] == 1: a
 * pivot of the normal equations was not safely positive (rank-deficient design, non-finite data); theta / weights
 * are then NOT written and the caller composes rlvi_wls_solve_f64 (minimum-norm solution, as the reference's
 * lstsq returns) / rlvi_linreg_losses_f64 / rlvi_update_weights_f64 itself, as for shapes beyond the limit.
 * ------------------------------------------------------------------------------------- */
int rlvi_linear_regression_check(int64_t n, int64_t d);
int rlvi_linear_regression_f64(const double *X, const double *y, int64_t n, int64_t d, int maxiter, double tol,
                               double estep_tol, int estep_maxiter, double *theta, double *weights,
                               int32_t *info, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------
 * One mini-batch of the online path in ONE launch, replaces online-learning/main.py:293-297:
 *   log_proba = log(0.5) for the first batch (first != 0; X, w may be NULL), else clf.predict_log_proba(X)[:, 1]
 *   = log sigmoid(X w + b) (:295, X.w on the fp64 matrix cores); residuals = -log_proba (:296 with :84-85: the
 *   target does not enter); sample_weight = update_weights_rlvi(residuals, tol, maxiter) (:297 -> :45-58).
 *   losses_out [n] (may be NULL) receives the residuals, out_iters (may be NULL) the iterations.  n <= 4096.
 * ------------------------------------------------------------------------------------- */
int rlvi_sample_weight_online_f64(const double *X, const double *w, double b, int first, int64_t n, int64_t d,
                                  double tol, int maxiter, double *losses_out, double *sample_weight,
                                  int32_t *out_iters, void *stream);

/* Measurement aid (bench.py: roofline.copy_same_bytes_us): flat copy, 16 B per lane, nontemporal loads and stores,
 * 16 waves per CU -- the plainest kernel that moves the M-step's bytes.  16-byte aligned, bytes a multiple of 16. */
int rlvi_stream_copy(void *dst, const void *src, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RLVI_HIP_H */
