"""ctypes/numpy front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the reported CPU baseline.  The product
(rlvi_amd/) never imports it and has no CPU fallback.

Parity pinning: tests/test_oracle_golden.py checks every function below against
fixtures generated from the imported reference (oracle/make_golden.py).

The fp32 deep-learning functions and the fp64 E-steps live in rlvi_oracle.c
(each citing its reference file:line).  The two standard-learning estimators
whose M-step is a third-party solver (scipy lstsq / sklearn liblinear) are
restated here in numpy around those solvers, as the reference does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librlvi_oracle.so")


def build(force=False):
    """Compile librlvi_oracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "rlvi_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "librlvi_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64 = ctypes.c_int64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.rlvi_oracle_num_threads.restype = ctypes.c_int
        L.rlvi_oracle_set_threads.argtypes = [ctypes.c_int]
        L.rlvi_oracle_nll_rows_f32.argtypes = [_f32p, _i64, _i64p, _i64, _i64, _f32p, _i32p]
        L.rlvi_oracle_label_rank_f32.argtypes = [_f32p, _i64, _i64p, _i64, _i64, _i32p]
        L.rlvi_oracle_label_rank_f32.restype = None
        L.rlvi_oracle_mstep_f32.restype = ctypes.c_int
        L.rlvi_oracle_mstep_f32.argtypes = [_f32p, _i64, _i64p, _i64p, _f32p, _f32p, _i64,
                                            _i64, _i64, _i64, _f32p, _i64, _f32p, _f32p, _f32p]
        L.rlvi_oracle_estep_deep_f32.restype = ctypes.c_int
        L.rlvi_oracle_estep_deep_f32.argtypes = [_f32p, _f32p, _i64, ctypes.c_float,
                                                 ctypes.c_int, _f32p, _f32p]
        L.rlvi_oracle_fn_threshold_f32.restype = ctypes.c_float
        L.rlvi_oracle_fn_threshold_f32.argtypes = [_f32p, _i64, ctypes.c_float, _i64p, _f32p]
        L.rlvi_oracle_truncate_f32.argtypes = [_f32p, _i64, ctypes.c_float, _u8p]
        L.rlvi_oracle_update_weights_f64.restype = ctypes.c_int
        L.rlvi_oracle_update_weights_f64.argtypes = [_f64p, _i64, ctypes.c_double,
                                                     ctypes.c_int, _f64p, _f64p]
        L.rlvi_oracle_update_weights_online_f64.restype = ctypes.c_int
        L.rlvi_oracle_update_weights_online_f64.argtypes = [_f64p, _i64, ctypes.c_double,
                                                            ctypes.c_int, _f64p]
        L.rlvi_oracle_linreg_losses_f64.restype = ctypes.c_double
        L.rlvi_oracle_linreg_losses_f64.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _i64, _f64p]
        L.rlvi_oracle_logistic_nll_f64.argtypes = [_f64p, _f64p, ctypes.c_double, _i64, _i64, _f64p]
        L.rlvi_oracle_linear_regression_f64.restype = ctypes.c_int
        L.rlvi_oracle_linear_regression_f64.argtypes = [_f64p, _f64p, _i64, _i64, ctypes.c_int, ctypes.c_double,
                                                        _f64p, _f64p, ctypes.POINTER(ctypes.c_int)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def num_threads():
    return lib().rlvi_oracle_num_threads()


def set_threads(n):
    lib().rlvi_oracle_set_threads(int(n))


# ---------------------------------------------------------------- deep (fp32)
def nll_rows(logits, labels):
    """a1+a6 -> (loss[B] f32, hit[B] i32).  train_rlvi.py:89, utils.py:65-79."""
    z = _c(logits, np.float32)
    y = _c(labels, np.int64)
    B, C = z.shape
    loss = np.empty(B, np.float32)
    hit = np.empty(B, np.int32)
    lib().rlvi_oracle_nll_rows_f32(_p(z, _f32p), C, _p(y, _i64p), B, C,
                                   _p(loss, _f32p), _p(hit, _i32p))
    return loss, hit


def accuracy(logits, labels, topk=(1,)):
    """deep-learning/utils.py:65-79: precision@k in per cent for every k of `topk` (list of floats); RuntimeError
    when max(topk) exceeds the number of classes, as torch.topk raises there."""
    z = _c(logits, np.float32)
    y = _c(labels, np.int64)
    B, C = z.shape
    if max(topk) > C:
        raise RuntimeError("selected index k out of range")
    rank = np.empty(B, np.int32)
    lib().rlvi_oracle_label_rank_f32(_p(z, _f32p), C, _p(y, _i64p), B, C, _p(rank, _i32p))
    return [100.0 * float((rank < k).sum()) / B for k in topk]


def mstep(logits, labels, idx, weights, residuals, scale_div=None, want_grad=True):
    """a1..a6 fused (train_rlvi.py:85-96 without the model).

    Mutates `residuals` (scatter).  Returns dict(loss, prec1, grad, loss_rows).
    """
    z = _c(logits, np.float32)
    y = _c(labels, np.int64)
    ix = _c(idx, np.int64)
    assert weights.dtype == np.float32 and residuals.dtype == np.float32
    assert residuals.flags.c_contiguous and weights.flags.c_contiguous
    B, C = z.shape
    N = weights.shape[0]
    grad = np.empty((B, C), np.float32) if want_grad else None
    rows = np.empty(B, np.float32)
    loss = ctypes.c_float()
    prec = ctypes.c_float()
    rc = lib().rlvi_oracle_mstep_f32(
        _p(z, _f32p), C, _p(y, _i64p), _p(ix, _i64p), _p(weights, _f32p),
        _p(residuals, _f32p), N, B, C, int(scale_div or B),
        _p(grad, _f32p) if want_grad else None, C, _p(rows, _f32p),
        ctypes.byref(loss), ctypes.byref(prec))
    if rc != 0:
        raise IndexError("label or index out of range")
    return dict(loss=np.float32(loss.value), prec1=np.float32(prec.value),
                grad=grad, loss_rows=rows)


def select_smallest(losses, k):
    """0/1 weights of np.argsort(loss)[:k] (train_usdnl.py:18-24, train_coteaching.py:18-30).
    Stable order: equal losses are taken by index (numpy's default sort leaves that open)."""
    mask = np.zeros(len(losses), np.float32)
    if k > 0:
        mask[np.argsort(losses, kind="stable")[:k]] = 1.0
    return mask


def _selected_ce(logits, labels, mask, div):
    B = logits.shape[0]
    r = mstep(logits, labels, np.arange(B, dtype=np.int64), mask, np.zeros(B, np.float32),
              scale_div=div)
    return r["loss"], r["grad"]


def usdnl_loss(logits, labels, forget_rate):
    """train_usdnl.loss_fn (:16-27): mean CE over the num_remember smallest losses -> (loss, grad)."""
    B = logits.shape[0]
    k = int((1 - forget_rate) * B)
    rows, _ = nll_rows(logits, labels)
    return _selected_ce(logits, labels, select_smallest(rows, k), k)


def coteaching_loss(y1, y2, t, forget_rate):
    """train_coteaching.loss_coteaching (:17-35): each model is averaged over the rows the OTHER
    one finds easy, and (as the reference does) divided by num_remember once more."""
    B = y1.shape[0]
    k = int((1 - forget_rate) * B)
    r1, _ = nll_rows(y1, t)
    r2, _ = nll_rows(y2, t)
    l1, g1 = _selected_ce(y1, t, select_smallest(r2, k), k * k)
    l2, g2 = _selected_ce(y2, t, select_smallest(r1, k), k * k)
    return l1, l2, g1, g2


def update_sample_weights(residuals, weights, tol=1e-3, maxiter=40, trace=False):
    """a7, in place on both arrays (train_rlvi.py:14-38).  Returns iterations
    (and the per-iteration error / mean-pi traces when trace=True)."""
    assert residuals.dtype == np.float32 and weights.dtype == np.float32
    assert residuals.flags.c_contiguous and weights.flags.c_contiguous
    N = residuals.shape[0]
    err = np.zeros(maxiter, np.float32)
    avg = np.zeros(maxiter, np.float32)
    it = lib().rlvi_oracle_estep_deep_f32(_p(residuals, _f32p), _p(weights, _f32p), N,
                                          np.float32(tol), int(maxiter),
                                          _p(err, _f32p), _p(avg, _f32p))
    if trace:
        return it, err[:it].copy(), avg[:it].copy()
    return it


def false_negative_criterion(weights, alpha=0.05, full=False):
    """a8 (train_rlvi.py:41-49) -> threshold f32 [, last_index, beta]."""
    w = _c(weights, np.float32)
    li = ctypes.c_int64()
    beta = ctypes.c_float()
    thr = lib().rlvi_oracle_fn_threshold_f32(_p(w, _f32p), w.shape[0], np.float32(alpha),
                                             ctypes.byref(li), ctypes.byref(beta))
    if full:
        return np.float32(thr), int(li.value), np.float32(beta.value)
    return np.float32(thr)


def truncate(weights, thr):
    """a9 in place (train_rlvi.py:103) + keep mask `weights > thr` (main.py:343)."""
    assert weights.dtype == np.float32 and weights.flags.c_contiguous
    mask = np.empty(weights.shape[0], np.uint8)
    lib().rlvi_oracle_truncate_f32(_p(weights, _f32p), weights.shape[0], np.float32(thr),
                                   _p(mask, _u8p))
    return mask.astype(bool)


def fn_margin(weights, alpha=0.05):
    """min_k |F_k - beta| / beta over the prefix sums: how far the
    threshold decision is from flipping by one position."""
    w = np.asarray(weights, np.float32)
    s = np.sort(w)[::-1]
    F = np.cumsum((np.float32(1) - s).astype(np.float64)).astype(np.float32)
    beta = np.float32(np.float32((np.float32(1) - w).astype(np.float64).sum()) * np.float32(alpha))
    if beta == 0:
        return 0.0
    return float(np.min(np.abs(F.astype(np.float64) - float(beta))) / float(beta))


# -------------------------------------------------------- standard / online (fp64)
def update_weights(losses, tol=1e-3, maxiter=100, trace=False):
    """a10 (standard-learning/rlvi.py:8-20)."""
    l = _c(losses, np.float64)
    out = np.empty_like(l)
    err = np.zeros(maxiter, np.float64)
    it = lib().rlvi_oracle_update_weights_f64(_p(l, _f64p), l.shape[0], float(tol),
                                              int(maxiter), _p(out, _f64p), _p(err, _f64p))
    if trace:
        return out, it, err[:it].copy()
    return out


def update_weights_rlvi(losses, tol=1e-3, maxiter=100, trace=False):
    """a13 (online-learning/main.py:45-58)."""
    l = _c(losses, np.float64)
    out = np.empty_like(l)
    it = lib().rlvi_oracle_update_weights_online_f64(_p(l, _f64p), l.shape[0], float(tol),
                                                     int(maxiter), _p(out, _f64p))
    if trace:
        return out, it
    return out


def linreg_losses(X, y, theta, w):
    """rlvi.py:72-74 / :81-83 -> (losses, sigma2)."""
    X = _c(X, np.float64)
    y = _c(y, np.float64)
    theta = _c(theta, np.float64)
    w = _c(w, np.float64)
    n, d = X.shape
    out = np.empty(n, np.float64)
    s2 = lib().rlvi_oracle_linreg_losses_f64(_p(X, _f64p), _p(y, _f64p), _p(theta, _f64p),
                                             _p(w, _f64p), n, d, _p(out, _f64p))
    return out, s2


def logistic_nll(X, w, b):
    """a14 (online-learning/main.py:295-296,:84-85): -log sigmoid(Xw+b)."""
    X = _c(X, np.float64)
    w = _c(w, np.float64)
    n, d = X.shape
    out = np.empty(n, np.float64)
    lib().rlvi_oracle_logistic_nll_f64(_p(X, _f64p), _p(w, _f64p), float(b), n, d, _p(out, _f64p))
    return out


def linear_regression(X, y, maxiter=100, tol=1e-3, trace=False):
    """a11 (standard-learning/rlvi.py:68-89).  The weighted least-squares solve
    is scipy.linalg.lstsq as in the reference, applied to sqrt(w)-scaled rows
    (the reference materialises diag(sqrt(w)); the product is the same matrix)."""
    from scipy.linalg import lstsq
    X = np.asarray(X, np.float64)
    y = np.asarray(y, np.float64)
    w = np.ones(X.shape[0])
    sw = np.sqrt(w)
    theta = lstsq(sw[:, None] * X, sw * y)[0]
    losses, _ = linreg_losses(X, y, theta, w)
    outer = 0
    for _ in range(maxiter):
        outer += 1
        w = update_weights(losses)
        prev = theta.copy()
        sw = np.sqrt(w)
        theta = lstsq(sw[:, None] * X, sw * y)[0]
        losses, _ = linreg_losses(X, y, theta, w)
        if np.linalg.norm(theta - prev) / np.linalg.norm(prev) <= tol:
            break
    if trace:
        return theta, w, outer
    return theta


def linear_regression_c(X, y, maxiter=100, tol=1e-3):
    """a11 as ONE C call (rlvi_oracle.c: Householder QR of the sqrt(w)-scaled rows in place of scipy's lstsq,
    full column rank only) -> (theta, w, outer, inner_total).  What bench.py times as cfg1's CPU baseline;
    pinned to the same golden G5 as `linear_regression` above.  Raises on a rank-deficient design."""
    X = _c(X, np.float64)
    y = _c(y, np.float64)
    n, d = X.shape
    theta = np.empty(d, np.float64)
    w = np.empty(n, np.float64)
    inner = ctypes.c_int(0)
    outer = lib().rlvi_oracle_linear_regression_f64(_p(X, _f64p), _p(y, _f64p), n, d, int(maxiter), float(tol),
                                                    _p(theta, _f64p), _p(w, _f64p), ctypes.byref(inner))
    if outer < 0:
        raise np.linalg.LinAlgError("rank-deficient design: use linear_regression (scipy lstsq, minimum norm)")
    return theta, w, outer, inner.value


# ---- estimators of SURVEY 8(f)-2, restated around the same E-step -------------------------
def update_weights_constrained(losses, n_eff, tol=1e-3, maxiter=100):
    """rlvi.py:23-43: unconstrained E-step, then the KKT shift if sum(w) < n_eff
    (the 1-D solve is scipy's minimize_scalar, as in the reference)."""
    from scipy import optimize as opt
    losses = np.asarray(losses, np.float64)
    n = len(losses)
    w = update_weights(losses, tol=tol, maxiter=maxiter)

    def shift_obj(s):
        return np.square(np.sum(np.exp(-losses + s) / ((n - n_eff) / n_eff + np.exp(-losses + s))) - n_eff)
    if np.sum(w) < n_eff:
        shift = opt.minimize_scalar(shift_obj)['x']
        w = np.exp(-losses + shift) / ((n - n_eff) / n_eff + np.exp(-losses + shift))
    return w


def _gauss_losses(sample, theta, w):
    r = np.linalg.norm(theta - sample, axis=1) ** 2
    return 0.5 * r / (w @ r / np.sum(w))


def mean(sample, maxiter=100, tol=1e-3):
    """rlvi.py:46-65."""
    sample = np.asarray(sample, np.float64)
    w = np.ones(sample.shape[0])
    theta = w @ sample / np.sum(w)
    losses = _gauss_losses(sample, theta, w)
    for _ in range(maxiter):
        w = update_weights(losses)
        prev = theta.copy()
        theta = w @ sample / np.sum(w)
        losses = _gauss_losses(sample, theta, w)
        if np.linalg.norm(theta - prev) / np.linalg.norm(prev) <= tol:
            break
    return theta


def _pca_step(sample, w):
    """utils.pca (standard-learning/utils.py:76-89): first principal axis of diag(w) @ samples
    (sklearn PCA centres the weighted rows), loss = |x|^2 - (x.theta)^2."""
    z = w[:, None] * sample
    z = z - z.mean(0)
    _, _, vt = np.linalg.svd(z, full_matrices=False)
    theta = vt[0] / np.linalg.norm(vt[0])
    # sklearn's svd_flip(u_based_decision=False): the largest-magnitude entry of the axis is > 0
    if theta[np.argmax(np.abs(theta))] < 0:
        theta = -theta
    losses = np.sum(sample ** 2, axis=1) - (sample @ theta) ** 2
    return theta, losses


def pca(sample, maxiter=100, tol=1e-2):
    """rlvi.py:111-125 (theta_init=None)."""
    sample = np.asarray(sample, np.float64)
    w = np.ones(sample.shape[0])
    theta, losses = _pca_step(sample, w)
    for _ in range(maxiter):
        w = update_weights(losses)
        prev = theta.copy()
        theta, losses = _pca_step(sample, w)
        if np.linalg.norm(theta - prev) / np.linalg.norm(prev) <= tol:
            break
    return theta


def _cov_step(sample, w):
    """utils.covariance (utils.py:92-108): weighted mean / covariance, Gaussian NLL per sample."""
    mu = sample.T @ w / np.sum(w)
    c = sample - mu
    cov = c.T @ (w[:, None] * c) / np.sum(w)
    sol = np.linalg.solve(cov, c.T)
    r = np.sum(c * sol.T, axis=1)
    sign, logdet = np.linalg.slogdet(cov)
    if sign <= 0:
        raise ValueError("Singular covariance matrix")
    return cov, 0.5 * (r + logdet + mu.shape[0] * np.log(2 * np.pi))


def covariance(sample, eps, maxiter=100, tol=1e-2):
    """rlvi.py:128-144."""
    sample = np.asarray(sample, np.float64)
    n = sample.shape[0]
    n_eff = n * (1 - eps)
    w = np.ones(n)
    theta, losses = _cov_step(sample, w)
    for _ in range(maxiter):
        w = update_weights_constrained(losses, n_eff)
        prev = theta.copy()
        theta, losses = _cov_step(sample, w)
        if np.linalg.norm(theta - prev, ord='fro') / np.linalg.norm(prev, ord='fro') <= tol:
            break
    return theta
