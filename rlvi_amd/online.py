"""MI355X mirror of the RLVI pieces of online-learning/main.py (numpy in, numpy out, fp64).

    update_weights_rlvi(losses, tol=1e-3, maxiter=100)    reference main.py:45-58
    cross_entropy(log_proba, targets)                     reference main.py:84-85
    residuals(X, coef, intercept)                         reference main.py:295-296 (log-sigmoid)
    rlvi_sample_weight(X, coef, intercept)                 the E-step inputs of main.py:291-299

The M-step itself is sklearn's SGDClassifier.partial_fit(sample_weight=...) in the reference
(sequential per-sample SGD inside a compiled third-party routine): out of scope, parity unpinned
(SURVEY 8(c)); this module produces the `sample_weight` it is fed.
"""
import numpy as np
import torch

from . import ops
from .standard import _dev


def update_weights_rlvi(losses, tol=1e-3, maxiter=100):
    '''Optimize Bernoulli probabilities (reference main.py:45-58).'''
    from .standard import _roundtrip_f64
    return _roundtrip_f64(losses, lambda l: ops.update_weights_f64(l, tol=tol, maxiter=maxiter, online=True)[0])


def cross_entropy(log_proba, targets):
    """reference main.py:84-85: -t*lp - (1-t)*lp, i.e. -log_proba whatever the target."""
    return -targets * log_proba - (1 - targets) * log_proba


def residuals(X, coef, intercept):
    """-log sigmoid(X @ coef + intercept): main.py:295-296 with cross_entropy folded in."""
    dev = _dev()
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
    wd = torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64).ravel()).to(dev)
    return ops.logistic_nll(Xd, wd, float(intercept)).cpu().numpy()


def rlvi_sample_weight(X, coef=None, intercept=0.0, tol=1e-3, maxiter=100):
    """sample_weight for clf.partial_fit (main.py:291-299).  coef None = classifier not fitted
    yet: log_proba = log(0.5) for every row (main.py:293).
    One launch (rlvi_sample_weight_online_f64: X.coef on the fp64 matrix cores -> -log sigmoid -> the online
    E-step) between one pinned H2D copy of [X | coef] and one pinned D2H copy; the host waits once.  Batches of
    more than 4096 rows take the two-launch composition."""
    from .standard import _STAGE
    dev = _dev()
    n = len(X)
    first = coef is None
    if n > 4096:
        if first:
            l = torch.full((n,), float(-np.log(0.5)), dtype=torch.float64, device=dev)
        else:
            Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
            wd = torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64).ravel()).to(dev)
            l = ops.logistic_nll(Xd, wd, float(intercept))
        w, _ = ops.update_weights_f64(l, tol=tol, maxiter=maxiter, online=True)
        return w.cpu().numpy()
    Xh = np.ascontiguousarray(X, dtype=np.float64)
    d = Xh.shape[1] if Xh.ndim == 2 else 1
    h_in, h_out, d_in, d_out = _STAGE.get(dev, "online", (n * d + d, n), device_side=True)
    if not first:
        h_in.numpy()[:n * d] = Xh.reshape(-1)
        h_in.numpy()[n * d:] = np.ascontiguousarray(coef, dtype=np.float64).ravel()
        d_in.copy_(h_in, non_blocking=True)
    ops.sample_weight_online(d_in[:n * d].view(n, d), d_in[n * d:], float(intercept), first=first,
                             tol=tol, maxiter=maxiter, out=d_out)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return h_out.numpy()[:n].copy()
