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


def rlvi_sample_weight(X, coef=None, intercept=0.0):
    """sample_weight for clf.partial_fit (main.py:291-299).  coef None = classifier not fitted
    yet: log_proba = log(0.5) for every row (main.py:293)."""
    dev = _dev()
    if coef is None:
        l = torch.full((len(X),), float(-np.log(0.5)), dtype=torch.float64, device=dev)
    else:
        Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
        wd = torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64).ravel()).to(dev)
        l = ops.logistic_nll(Xd, wd, float(intercept))
    w, _ = ops.update_weights_f64(l, online=True)
    return w.cpu().numpy()
