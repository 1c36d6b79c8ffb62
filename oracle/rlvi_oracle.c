/*
 * rlvi_oracle.c -- CPU restatement of the RLVI E-step / M-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product path (rlvi_amd/) never falls back to it.
 *
 * Parity pinning: every function here is checked against golden vectors
 * produced by importing the reference itself (oracle/make_golden.py, run in
 * the build container; fixtures under tests/golden/).  See
 * tests/test_oracle_golden.py.
 *
 * Each function cites the reference lines (relative to /root/reference) that
 * it restates.  The arithmetic type follows the reference: fp32 for the
 * deep-learning path (torch CPU float32), fp64 for the numpy paths.
 * Reductions (sum / mean / norm) are accumulated in fp64 and rounded once;
 * torch's fp32 cascade sums land within a few ulp of that.
 *
 * Build: see oracle/Makefile   (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int rlvi_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void rlvi_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- *
 * a1 + a6: per-sample NLL and top-1 hit.
 *   deep-learning/methods/train_rlvi.py:89  F.cross_entropy(reduction='none')
 *   deep-learning/utils.py:65-79            accuracy(): softmax -> topk -> eq
 * log-softmax is evaluated the way torch does it: (z - max) - log(sum exp).
 * Top-1 hit: the label is the FIRST column that attains the row maximum -- exactly one
 * column of a row can be a hit.  That is the order of torch.max / argmax, which the
 * reference's evaluate() uses (deep-learning/utils.py:57-58); accuracy()'s topk picks one of
 * several tied columns in an unspecified order (utils.py:70), so for train_acc the WHICH is
 * "parity unpinned", the HOW MANY (one) is not.  The kernels use the same rule as here.
 * (Both reference paths look at softmax(logits); columns whose logits differ but whose
 *  fp32 softmax values coincide are not reproduced.)
 * ------------------------------------------------------------------------- */
static int first_max_is(const float *z, int64_t C, int64_t y, float m) {
    (void)C;
    if (z[y] != m) return 0;
    for (int64_t c = 0; c < y; ++c)
        if (z[c] == m) return 0;
    return 1;
}

static inline void row_stats_f32(const float *z, int64_t C, float *m_out,
                                 float *logs_out, int64_t *amax_out) {
    float m = z[0];
    int64_t am = 0;
    for (int64_t c = 1; c < C; ++c)
        if (z[c] > m) { m = z[c]; am = c; }
    double s = 0.0;
    for (int64_t c = 0; c < C; ++c) s += (double)expf(z[c] - m);
    *m_out = m;
    *logs_out = logf((float)s);
    *amax_out = am;
}

void rlvi_oracle_nll_rows_f32(const float *logits, int64_t ld,
                              const int64_t *labels, int64_t B, int64_t C,
                              float *loss, int32_t *hit) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        const float *z = logits + i * ld;
        float m, logs;
        int64_t am;
        row_stats_f32(z, C, &m, &logs, &am);
        const int64_t y = labels[i];
        loss[i] = -((z[y] - m) - logs);
        if (hit) hit[i] = first_max_is(z, C, y, m);
    }
}

/* ------------------------------------------------------------------------- *
 * precision@k: accuracy(logit, target, topk) of deep-learning/utils.py:65-79.
 * The reference ranks the SOFTMAX values with torch.topk (:70) and counts the
 * rows whose label is among the first k (:72, :76-77).  Restated on the logits:
 * rank_i = #{c : z_ic > z_iy} + #{c < y : z_ic == z_iy} (equal values in column
 * order; which of several equal values topk lists first is an implementation
 * detail of the reference, and so are logits that differ but whose fp32 softmax
 * values coincide -- unpinned, as for top-1 in G9's note), hit@k = rank_i < k.
 * A label outside [0, C) matches no prediction (:72): never a hit.
 * ------------------------------------------------------------------------- */
void rlvi_oracle_label_rank_f32(const float *logits, int64_t ld,
                                const int64_t *labels, int64_t B, int64_t C,
                                int32_t *rank) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        const float *z = logits + i * ld;
        const int64_t y = labels[i];
        if (y < 0 || y >= C) { rank[i] = (int32_t)C; continue; }
        int32_t r = 0;
        for (int64_t c = 0; c < C; ++c)
            r += (z[c] > z[y] || (z[c] == z[y] && c < y)) ? 1 : 0;
        rank[i] = r;
    }
}

/* ------------------------------------------------------------------------- *
 * a1..a6 fused: one mini-batch of the M-step with lagged pi.
 *   train_rlvi.py:85     prec = accuracy(logits, labels)      -> *prec1
 *   train_rlvi.py:89     loss_i = CE(logits_i, y_i)
 *   train_rlvi.py:90     residuals[indexes] = loss
 *   train_rlvi.py:92-94  L = mean(loss * weights[indexes])    -> *loss_mean
 *   train_rlvi.py:96     dL/dlogits = (pi_i/B)(softmax - onehot) -> grad
 * `scale_div` is the divisor of the mean (B for one device; the GLOBAL batch
 * when the batch is sharded, see rlvi_amd/dist.py).
 * Returns 0, or -1 on an out-of-range label / index.
 * ------------------------------------------------------------------------- */
int rlvi_oracle_mstep_f32(const float *logits, int64_t ld,
                          const int64_t *labels, const int64_t *idx,
                          const float *weights, float *residuals, int64_t N,
                          int64_t B, int64_t C, int64_t scale_div,
                          float *grad, int64_t ldg, float *loss_rows,
                          float *loss_mean, float *prec1) {
    for (int64_t i = 0; i < B; ++i) {
        if (labels[i] < 0 || labels[i] >= C) return -1;
        if (idx[i] < 0 || idx[i] >= N) return -1;
    }
    const float invB = 1.0f / (float)scale_div;
    float *lrow = loss_rows ? loss_rows : (float *)malloc((size_t)B * sizeof(float));
    double acc = 0.0;
    int64_t hits = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc, hits)
    for (int64_t i = 0; i < B; ++i) {
        const float *z = logits + i * ld;
        float m, logs;
        int64_t am;
        row_stats_f32(z, C, &m, &logs, &am);
        const int64_t y = labels[i];
        const float li = -((z[y] - m) - logs);
        const float pi = weights[idx[i]];      /* :92  gather (lagged pi) */
        lrow[i] = li;
        acc += (double)(li * pi);              /* :93 */
        hits += first_max_is(z, C, y, m);
        if (grad) {
            const float g = pi * invB;
            float *gr = grad + i * ldg;
            for (int64_t c = 0; c < C; ++c) {
                const float p = expf((z[c] - m) - logs);
                gr[c] = p * g - (c == y ? g : 0.0f);
            }
        }
    }
    for (int64_t i = 0; i < B; ++i) residuals[idx[i]] = lrow[i];   /* :90 */
    if (!loss_rows) free(lrow);
    *loss_mean = (float)(acc / (double)scale_div);
    *prec1 = (float)hits * (float)(100.0 / (double)B);
    return 0;
}

/* ------------------------------------------------------------------------- *
 * a7: deep-learning E-step, in place.
 *   train_rlvi.py:27  residuals.sub_(residuals.min())
 *   train_rlvi.py:28  exp_res = exp(-residuals)
 *   train_rlvi.py:29  avg = 0.95
 *   train_rlvi.py:31  ratio = avg / (1 - avg)
 *   train_rlvi.py:32  new = ratio*e / (1 + ratio*e)
 *   train_rlvi.py:33  error = ||new - weights||_2   (iteration 1: caller's pi)
 *   train_rlvi.py:34-35  weights[:] = new ; avg = mean(weights)
 *   train_rlvi.py:36  stop if error < tol   (after the assignment)
 *   train_rlvi.py:38  weights /= max(weights)
 * Returns the number of iterations executed.  err_trace / avg_trace (may be
 * NULL) receive the per-iteration error and mean-pi (length >= maxiter).
 * Iteration 1 uses the python-float ratio 0.95/(1-0.95) cast to fp32 (=19.0f);
 * later iterations do avg/(1-avg) in fp32, exactly as the 0-dim tensors do.
 * ------------------------------------------------------------------------- */
int rlvi_oracle_estep_deep_f32(float *residuals, float *weights, int64_t N,
                               float tol, int maxiter, float *err_trace,
                               float *avg_trace) {
    if (N <= 0) return 0;
    float mn = residuals[0];
    for (int64_t i = 1; i < N; ++i)
        if (residuals[i] < mn) mn = residuals[i];
    float *e = (float *)malloc((size_t)N * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        residuals[i] = residuals[i] - mn;
        e[i] = expf(-residuals[i]);
    }
    float ratio = (float)(0.95 / (1.0 - 0.95));
    int it = 0;
    for (it = 0; it < maxiter;) {
        double sse = 0.0, sum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sse, sum)
        for (int64_t i = 0; i < N; ++i) {
            const float t = ratio * e[i];
            const float nw = t / (1.0f + t);
            const float d = nw - weights[i];
            sse += (double)(d * d);
            sum += (double)nw;
            weights[i] = nw;
        }
        const float err = (float)sqrt(sse);
        const float avg = (float)sum / (float)N;
        if (err_trace) err_trace[it] = err;
        if (avg_trace) avg_trace[it] = avg;
        ++it;
        if (err < tol) break;
        ratio = avg / (1.0f - avg);
    }
    float mx = weights[0];
    for (int64_t i = 1; i < N; ++i)
        if (weights[i] > mx) mx = weights[i];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) weights[i] = weights[i] / mx;
    free(e);
    return it;
}

/* ------------------------------------------------------------------------- *
 * a8: type-II-error threshold.
 *   train_rlvi.py:43-44  beta = alpha * sum(1 - w)
 *   train_rlvi.py:45     sorted descending
 *   train_rlvi.py:46     F = cumsum(1 - sorted)   (torch CPU: fp64 running
 *                        sum, every prefix rounded to fp32 -- SURVEY 7.2-3)
 *   train_rlvi.py:47     last_index = #{F <= beta} - 1   (-1 wraps to N-1)
 *   train_rlvi.py:48     threshold = sorted[last_index]
 * ------------------------------------------------------------------------- */
static int cmp_desc_f32(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

float rlvi_oracle_fn_threshold_f32(const float *weights, int64_t N, float alpha,
                                   int64_t *last_index_out, float *beta_out) {
    float *s = (float *)malloc((size_t)N * sizeof(float));
    memcpy(s, weights, (size_t)N * sizeof(float));
    double tot = 0.0;
    for (int64_t i = 0; i < N; ++i) tot += (double)(1.0f - s[i]);
    const float beta = (float)tot * alpha;
    qsort(s, (size_t)N, sizeof(float), cmp_desc_f32);
    double run = 0.0;
    int64_t count = 0;
    for (int64_t i = 0; i < N; ++i) {
        run += (double)(1.0f - s[i]);
        if ((float)run <= beta) ++count;
    }
    int64_t li = count - 1;
    if (li < 0) li += N;
    const float thr = s[li];
    if (last_index_out) *last_index_out = count - 1;
    if (beta_out) *beta_out = beta;
    free(s);
    return thr;
}

/* ------------------------------------------------------------------------- *
 * a9: truncation + selection mask.
 *   train_rlvi.py:103   weights[weights < threshold] = 0
 *   main.py:343         mask = (sample_weights > threshold)
 * ------------------------------------------------------------------------- */
void rlvi_oracle_truncate_f32(float *weights, int64_t N, float thr,
                              uint8_t *mask_gt) {
    for (int64_t i = 0; i < N; ++i) {
        if (weights[i] < thr) weights[i] = 0.0f;
        if (mask_gt) mask_gt[i] = weights[i] > thr;
    }
}

/* ------------------------------------------------------------------------- *
 * a10: numpy E-step of the standard-learning path (fp64).
 *   standard-learning/rlvi.py:10   w = 0.95
 *   rlvi.py:13-14   eps = 1 - mean(w); ratio = eps/(1-eps)
 *   rlvi.py:15      new = exp(-l)/(ratio + exp(-l))
 *   rlvi.py:16-19   error = ||new - w||; w = new; stop if error < tol
 * Returns iterations.
 * ------------------------------------------------------------------------- */
int rlvi_oracle_update_weights_f64(const double *losses, int64_t n, double tol,
                                   int maxiter, double *out, double *err_trace) {
    double *w = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) { w[i] = 0.95; out[i] = 0.95; }
    int it = 0;
    for (it = 0; it < maxiter;) {
        double sum = 0.0;
        for (int64_t i = 0; i < n; ++i) sum += w[i];
        const double eps = 1.0 - sum / (double)n;
        const double ratio = eps / (1.0 - eps);
        double sse = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double e = exp(-losses[i]);
            const double nw = e / (ratio + e);
            const double d = nw - w[i];
            sse += d * d;
            out[i] = nw;
        }
        const double err = sqrt(sse);
        if (err_trace) err_trace[it] = err;
        memcpy(w, out, (size_t)n * sizeof(double));
        ++it;
        if (err < tol) break;
    }
    free(w);
    return it;
}

/* ------------------------------------------------------------------------- *
 * a13: online E-step (fp64).
 *   online-learning/main.py:47-48  e = exp(-l); w = 0.5
 *   main.py:50-52   avg = mean(w); ratio = avg/(1-avg); new = ratio e/(1+ratio e)
 *   main.py:53-56   error = ||new - w||; break BEFORE assigning if error < tol
 *   main.py:57      new /= max(new) * len(new)
 * Returns iterations (number of times `new` was evaluated).
 * ------------------------------------------------------------------------- */
int rlvi_oracle_update_weights_online_f64(const double *losses, int64_t n,
                                          double tol, int maxiter, double *out) {
    double *w = (double *)malloc((size_t)n * sizeof(double));
    double *e = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) { w[i] = 0.5; e[i] = exp(-losses[i]); out[i] = 0.5; }
    int it = 0;
    for (it = 0; it < maxiter;) {
        double sum = 0.0;
        for (int64_t i = 0; i < n; ++i) sum += w[i];
        const double avg = sum / (double)n;
        const double ratio = avg / (1.0 - avg);
        double sse = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double t = ratio * e[i];
            const double nw = t / (1.0 + t);
            const double d = nw - w[i];
            sse += d * d;
            out[i] = nw;
        }
        ++it;
        if (sqrt(sse) < tol) break;
        memcpy(w, out, (size_t)n * sizeof(double));
    }
    double mx = out[0];
    for (int64_t i = 1; i < n; ++i)
        if (out[i] > mx) mx = out[i];
    const double den = mx * (double)n;
    for (int64_t i = 0; i < n; ++i) out[i] = out[i] / den;
    free(w);
    free(e);
    return it;
}

/* ------------------------------------------------------------------------- *
 * a11 pieces: Gaussian NLL of the linear-regression path (fp64).
 *   rlvi.py:72/81   residuals = (y - X @ theta)**2
 *   rlvi.py:73/82   sigma2 = w @ residuals / sum(w)
 *   rlvi.py:74/83   losses = 0.5 * residuals / sigma2
 * X is row-major [n, d].  Returns sigma2.
 * ------------------------------------------------------------------------- */
double rlvi_oracle_linreg_losses_f64(const double *X, const double *y,
                                     const double *theta, const double *w,
                                     int64_t n, int64_t d, double *losses) {
    double num = 0.0, den = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double p = 0.0;
        for (int64_t j = 0; j < d; ++j) p += X[i * d + j] * theta[j];
        const double r = (y[i] - p) * (y[i] - p);
        losses[i] = r;
        num += w[i] * r;
        den += w[i];
    }
    const double sigma2 = num / den;
    for (int64_t i = 0; i < n; ++i) losses[i] = 0.5 * losses[i] / sigma2;
    return sigma2;
}

/* ------------------------------------------------------------------------- *
 * a14: online residual  l_i = -log sigmoid(x_i . w + b)   (fp64)
 *   online-learning/main.py:295-296, :84-85 (target-independent)
 * ------------------------------------------------------------------------- */
void rlvi_oracle_logistic_nll_f64(const double *X, const double *wv, double b,
                                  int64_t n, int64_t d, double *losses) {
    for (int64_t i = 0; i < n; ++i) {
        double p = b;
        for (int64_t j = 0; j < d; ++j) p += X[i * d + j] * wv[j];
        /* -log sigma(p) = log1p(exp(-p)) for p >= 0, -p + log1p(exp(p)) else */
        losses[i] = p >= 0.0 ? log1p(exp(-p)) : -p + log1p(exp(p));
    }
}

/* ------------------------------------------------------------------------- *
 * a11: the whole estimator  linear_regression(X, y, maxiter=100, tol=1e-3)   (fp64)
 *   standard-learning/rlvi.py:69-71  weights = 1; theta = lstsq(diag(sqrt(w)) X, diag(sqrt(w)) y)
 *   rlvi.py:72-74                    residuals, sigma2, losses (rlvi_oracle_linreg_losses_f64)
 *   rlvi.py:76-87                    repeat: weights = update_weights(losses) (tol 1e-3, maxiter 100: :8-20);
 *                                    theta_prev = theta; theta = lstsq(...); losses; stop when
 *                                    ||theta - prev|| / ||prev|| <= tol
 * The reference's least-squares solve is scipy.linalg.lstsq (LAPACK gelsd, third party) on the sqrt(w)-scaled
 * rows; here the same least-squares problem is solved by Householder QR of the scaled rows (no normal
 * equations: the conditioning of the reference's solve).  Full column rank only -- a rank-deficient design
 * (gelsd: minimum-norm solution) returns -1 and is left to the numpy restatement in rlvi_oracle.py.
 * Pinned by tests/golden/g5_standard.npz (theta, final weights, outer-iteration count of the reference).
 * Returns the number of outer iterations; inner_total (may be NULL) receives the E-step iterations in all.
 * ------------------------------------------------------------------------- */
static int oracle_wls_qr(const double *X, const double *y, const double *w, int64_t n, int64_t d,
                         double *A, double *b, double *theta) {
    /* A = diag(sqrt(w)) X (n x d, row-major), b = sqrt(w) y */
    for (int64_t i = 0; i < n; ++i) {
        const double s = sqrt(w[i]);
        for (int64_t j = 0; j < d; ++j) A[i * d + j] = s * X[i * d + j];
        b[i] = s * y[i];
    }
    double amax = 0.0;
    for (int64_t k = 0; k < d; ++k) {
        double nrm2 = 0.0;
        for (int64_t i = k; i < n; ++i) nrm2 += A[i * d + k] * A[i * d + k];
        const double nrm = sqrt(nrm2);
        if (k == 0 || nrm > amax) amax = nrm > amax ? nrm : amax;
        if (!(nrm > amax * (double)d * 64.0 * 2.220446049250313e-16)) return -1;     /* rank-deficient / not finite */
        const double alpha = A[k * d + k] > 0.0 ? -nrm : nrm;
        /* v = x - alpha e_k (stored in place), H = I - 2 v v^T / (v^T v) */
        const double vkk = A[k * d + k] - alpha;
        double vtv = vkk * vkk;
        for (int64_t i = k + 1; i < n; ++i) vtv += A[i * d + k] * A[i * d + k];
        A[k * d + k] = vkk;
        for (int64_t j = k + 1; j < d; ++j) {
            double dot = 0.0;
            for (int64_t i = k; i < n; ++i) dot += A[i * d + k] * A[i * d + j];
            const double f = 2.0 * dot / vtv;
            for (int64_t i = k; i < n; ++i) A[i * d + j] -= f * A[i * d + k];
        }
        {
            double dot = 0.0;
            for (int64_t i = k; i < n; ++i) dot += A[i * d + k] * b[i];
            const double f = 2.0 * dot / vtv;
            for (int64_t i = k; i < n; ++i) b[i] -= f * A[i * d + k];
        }
        A[k * d + k] = alpha;                 /* R's diagonal; the rows below hold v and are not read again */
    }
    for (int64_t k = d - 1; k >= 0; --k) {    /* R theta = (Q^T b)[0 .. d) */
        double s = b[k];
        for (int64_t j = k + 1; j < d; ++j) s -= A[k * d + j] * theta[j];
        theta[k] = s / A[k * d + k];
    }
    return 0;
}

int rlvi_oracle_linear_regression_f64(const double *X, const double *y, int64_t n, int64_t d, int maxiter,
                                      double tol, double *theta, double *w, int *inner_total) {
    double *A = (double *)malloc((size_t)n * (size_t)d * sizeof(double));
    double *b = (double *)malloc((size_t)n * sizeof(double));
    double *losses = (double *)malloc((size_t)n * sizeof(double));
    double *prev = (double *)malloc((size_t)d * sizeof(double));
    double *wn = (double *)malloc((size_t)n * sizeof(double));
    int outer = 0, inner = 0, rc = 0;
    for (int64_t i = 0; i < n; ++i) w[i] = 1.0;
    rc = oracle_wls_qr(X, y, w, n, d, A, b, theta);
    if (rc == 0) {
        rlvi_oracle_linreg_losses_f64(X, y, theta, w, n, d, losses);
        for (int o = 0; o < maxiter; ++o) {
            ++outer;
            inner += rlvi_oracle_update_weights_f64(losses, n, 1e-3, 100, wn, NULL);
            memcpy(w, wn, (size_t)n * sizeof(double));
            memcpy(prev, theta, (size_t)d * sizeof(double));
            rc = oracle_wls_qr(X, y, w, n, d, A, b, theta);
            if (rc != 0) break;
            rlvi_oracle_linreg_losses_f64(X, y, theta, w, n, d, losses);
            double dn = 0.0, pn = 0.0;
            for (int64_t j = 0; j < d; ++j) {
                dn += (theta[j] - prev[j]) * (theta[j] - prev[j]);
                pn += prev[j] * prev[j];
            }
            if (sqrt(dn) / sqrt(pn) <= tol) break;
        }
    }
    if (inner_total) *inner_total = inner;
    free(A); free(b); free(losses); free(prev); free(wn);
    return rc != 0 ? -1 : outer;
}
