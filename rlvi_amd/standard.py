"""MI355X mirror of standard-learning/rlvi.py (numpy in, numpy out, fp64).

    update_weights(losses, tol=1e-3, maxiter=100)             reference rlvi.py:8-20
    update_weights_constrained(losses, n_eff, ...)            reference rlvi.py:23-43
    mean / pca / covariance                                    reference rlvi.py:46-65, :111-144
    linear_regression(X, y, maxiter=100, tol=1e-3) -> theta    reference rlvi.py:68-89
    logistic_regression(X, y, maxiter=100, tol=1e-2) -> theta  reference rlvi.py:92-108

The E-step fixed point and the per-sample NLL (the X.theta contraction) run in
librlvi_gfx950.so, and so does the weighted least-squares solve of linear_regression (normal
equations on the fp64 MFMA units + Cholesky); logistic regression keeps sklearn's liblinear on
the host, as the reference has it.
Host arrays cross PCIe once per call; use rlvi_amd.ops directly to keep data resident.
"""
import numpy as np
import torch

from . import _lib, ops


def _dev():
    if not torch.cuda.is_available():
        from ._lib import RlviError
        raise RlviError("rlvi_amd.standard needs an MI355X HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class _Staging:
    """Pinned host buffers for the numpy-in / numpy-out calls: one set per (thread, device, tag, sizes), least
    recently used evicted beyond 32 sets.  Keyed by the thread as well: the buffers are written, copied
    asynchronously and read back around ONE host wait, so two threads must never share a set."""

    def __init__(self, cap=32):
        import collections
        import threading
        self._lock = threading.Lock()
        self._sets = collections.OrderedDict()
        self._cap = cap

    def get(self, dev, tag, sizes, dtype=torch.float64, device_side=False):
        """(h_0, h_1, ...) pinned host buffers of the given lengths; device_side: followed by device buffers of the
        same lengths (kept with the set: a call then allocates nothing)."""
        import threading
        key = (threading.get_ident(), dev.index, tag, tuple(int(x) for x in sizes), dtype, device_side)
        with self._lock:
            st = self._sets.get(key)
            if st is not None:
                self._sets.move_to_end(key)
                return st
        st = tuple(torch.empty(max(int(x), 1), dtype=dtype).pin_memory() for x in sizes)
        if device_side:
            st = st + tuple(torch.empty(max(int(x), 1), dtype=dtype, device=dev) for x in sizes)
        with self._lock:
            self._sets[key] = st
            while len(self._sets) > self._cap:
                self._sets.popitem(last=False)
        return st


_STAGE = _Staging()


def _roundtrip_f64(host_in, fn):
    """numpy in, numpy out through pinned staging buffers: both copies are queued on the stream around the
    launch and the host waits ONCE -- a pageable `.to(device)` and a `.cpu()` are two synchronising copies of
    their own (update_weights at n = 40 ... 1000: 60-66 -> 51-54 us per call; the fp64 iterative kernel and four
    torch calls are the rest).  The result has the input's shape."""
    dev = _dev()
    a = np.ascontiguousarray(host_in, dtype=np.float64)
    shape = a.shape
    a = a.reshape(-1)
    if a.size == 0:
        return np.empty(shape, np.float64)
    h_in, h_out = _STAGE.get(dev, "roundtrip", (a.size, a.size))
    h_in.numpy()[:] = a
    d_out = fn(h_in.to(dev, non_blocking=True))
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return h_out.numpy().copy().reshape(shape)


def update_weights(losses, tol=1e-3, maxiter=100):
    '''Optimize Bernoulli probabilities (reference rlvi.py:8-20).'''
    return _roundtrip_f64(losses, lambda l: ops.update_weights_f64(l, tol=tol, maxiter=maxiter)[0])


def _wls(X, y, w):
    """theta = argmin sum_i w_i (y_i - x_i.theta)^2 (rlvi.py:70-71,79-80; the reference materialises
    the n x n diag(sqrt(w)) and calls scipy lstsq).  Here: the weighted Gram matrix on the fp64
    matrix cores + a Cholesky solve, all on the device (rlvi_wls_solve_f64); d > 63 falls back to
    torch.linalg.lstsq on the sqrt(w)-scaled rows."""
    if X.shape[1] <= 63:
        return ops.wls_solve(X, y, w)
    sw = torch.sqrt(w)
    sol = torch.linalg.lstsq(sw[:, None] * X, (sw * y)[:, None])
    return sol.solution[:, 0]


def _linear_regression_host_loop(X, y, maxiter, tol):
    """The general path (any n, d <= 63 on the device solver, rank-deficient designs through the minimum-norm
    kernel): one launch per statement of rlvi.py:76-87 and the stop test on the host, one sync per outer
    iteration.  X, y: device tensors."""
    dev = X.device
    w = torch.ones(X.shape[0], dtype=torch.float64, device=dev)
    theta = _wls(X, y, w)
    losses, _ = ops.linreg_losses(X, y, theta, w)          # rlvi.py:72-74
    outer = 0
    for _ in range(maxiter):
        outer += 1
        w, _ = ops.update_weights_f64(losses)               # rlvi.py:77
        prev = theta
        theta = _wls(X, y, w)                               # rlvi.py:79-80
        losses, _ = ops.linreg_losses(X, y, theta, w)       # rlvi.py:81-83
        disc = torch.linalg.norm(theta - prev) / torch.linalg.norm(prev)
        if bool(disc <= tol):                               # rlvi.py:85-87 (one host sync)
            break
    return theta, w, outer


def linear_regression(X, y, maxiter=100, tol=1e-3, return_info=False):
    """rlvi.py:68-89.  n <= 4096, d <= 31 (the reference's 40 x 10, BASELINE's 1000 x 20): the whole estimator is
    ONE launch (rlvi_linear_regression_f64: E-step, weighted least squares on the fp64 matrix cores, NLL and
    the stop test ||theta - prev|| / ||prev|| <= tol all on the device) between one pinned H2D copy of [X | y]
    and one pinned D2H copy of {theta, weights, info}; the host waits once.  Beyond that, or when the launch
    reports a rank-deficient design: the general path."""
    dev = _dev()
    Xh = np.ascontiguousarray(X, dtype=np.float64)
    yh = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    n, d = Xh.shape
    ws = ops.workspace(dev, n, 0)
    if n > 0 and d > 0 and ops.linear_regression_check(n, d):
        h_in, h_out, d_in, d_out = _STAGE.get(dev, "linreg", (n * d + n, d + n + 2), device_side=True)
        h_in.numpy()[:n * d] = Xh.reshape(-1)
        h_in.numpy()[n * d:] = yh
        d_in.copy_(h_in, non_blocking=True)
        info = d_out[d + n:].view(torch.int32)              # 4 x int32 behind theta and the weights
        ops.linear_regression(d_in[:n * d].view(n, d), d_in[n * d:], maxiter=maxiter, tol=tol,
                              theta=d_out[:d], weights=d_out[d:d + n], info=info, ws=ws)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()        # the one host wait of the call
        res = h_out.numpy()
        inf = res[d + n:].view(np.int32)
        if inf[3] == 0:
            if return_info:
                return res[:d].copy(), res[d:d + n].copy(), int(inf[0])
            return res[:d].copy()
        Xd, yd = d_in[:n * d].view(n, d), d_in[n * d:]
    else:
        Xd, yd = torch.from_numpy(Xh).to(dev), torch.from_numpy(yh).to(dev)
    theta, w, outer = _linear_regression_host_loop(Xd, yd, maxiter, tol)
    th = theta.cpu().numpy()
    if not np.isfinite(th).all():
        # the reference's error behaviour: scipy's lstsq (rlvi.py:71,80; check_finite) raises this ValueError as
        # soon as a non-finite number reaches it -- in X or y, or in the weights of an exactly interpolating fit
        # (variance 0 -> 0/0 losses); here the NaN has run through to theta instead, which happens in no other case
        ws.clear_status()
        raise ValueError("array must not contain infs or NaNs")
    # a cooperating launch that could not run, a fixed point that did not converge: never silent
    # (RLVI_ST_SINGULAR is information: the minimum-norm solution was returned, as lstsq does)
    ws.raise_on_status("linear_regression", mask=_lib.ST_TIMEOUT | _lib.ST_NOCONV)
    if return_info:
        return th, w.cpu().numpy(), outer
    return th


def _sklearn_log_reg(X_host, y_host, X_dev, w_dev, reg_coeff=1e2):
    """utils.sklearn_log_reg (standard-learning/utils.py:61-73): liblinear fit on the host (third
    party), per-sample loss -log p(class 0 | x) on the device (it does not depend on y)."""
    from sklearn.linear_model import LogisticRegression
    w = (w_dev / w_dev.max()).cpu().numpy()                 # utils.py:66
    clf = LogisticRegression(solver="liblinear", C=reg_coeff)
    clf.fit(X_host, y_host, sample_weight=w)
    theta = np.hstack([clf.intercept_.flatten(), clf.coef_.flatten()])
    coef = torch.from_numpy(clf.coef_.flatten().copy()).to(X_dev.device)
    # -log p(class 0) = -log sigmoid(-(x.w + b)) = logistic_nll(X, -w, -b)
    losses = ops.logistic_nll(X_dev, -coef, -float(clf.intercept_[0]))
    return theta, losses


def logistic_regression(X, y, maxiter=100, tol=1e-2):
    dev = _dev()
    Xh = np.ascontiguousarray(X, dtype=np.float64)
    yh = np.asarray(y)
    Xd = torch.from_numpy(Xh).to(dev)
    w = torch.ones(Xh.shape[0], dtype=torch.float64, device=dev)
    theta, losses = _sklearn_log_reg(Xh, yh, Xd, w)
    for _ in range(maxiter):
        w, _ = ops.update_weights_f64(losses)
        prev = theta.copy()
        theta, losses = _sklearn_log_reg(Xh, yh, Xd, w)
        if np.linalg.norm(theta - prev) / np.linalg.norm(prev) <= tol:
            break
    return theta


# ---- the remaining estimators of standard-learning/rlvi.py on the same GPU E-step.
# n is tens to hundreds here: the E-step runs in librlvi_gfx950.so, the small dense algebra around
# it (weighted means, a 2x2 SVD / solve, scipy's 1-D KKT solve) stays on the host as numpy, exactly
# where the reference has it.
def update_weights_constrained(losses, n_eff, tol=1e-3, maxiter=100):
    """rlvi.py:23-43."""
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
    """utils.pca (standard-learning/utils.py:76-89) with the PCA fit as an SVD of the centred,
    weight-scaled rows and sklearn's sign convention."""
    z = w[:, None] * sample
    z = z - z.mean(0)
    _, _, vt = np.linalg.svd(z, full_matrices=False)
    theta = vt[0] / np.linalg.norm(vt[0])
    if theta[np.argmax(np.abs(theta))] < 0:      # sklearn's svd_flip(u_based_decision=False)
        theta = -theta
    return theta, np.sum(sample ** 2, axis=1) - (sample @ theta) ** 2


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
    """utils.covariance (utils.py:92-108)."""
    mu = sample.T @ w / np.sum(w)
    c = sample - mu
    cov = c.T @ (w[:, None] * c) / np.sum(w)
    r = np.sum(c * np.linalg.solve(cov, c.T).T, axis=1)
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
