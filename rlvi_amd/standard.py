"""MI355X mirror of standard-learning/rlvi.py (numpy in, numpy out, fp64).

    update_weights(losses, tol=1e-3, maxiter=100)             reference rlvi.py:8-20
    linear_regression(X, y, maxiter=100, tol=1e-3) -> theta    reference rlvi.py:68-89
    logistic_regression(X, y, maxiter=100, tol=1e-2) -> theta  reference rlvi.py:92-108

The E-step fixed point and the per-sample NLL (the X.theta contraction) run in
librlvi_gfx950.so; the weighted solvers stay third-party as in the reference (scipy lstsq /
sklearn liblinear there; torch.linalg.lstsq on the device / sklearn liblinear here).
Host arrays cross PCIe once per call; use rlvi_amd.ops directly to keep data resident.
"""
import numpy as np
import torch

from . import ops


def _dev():
    if not torch.cuda.is_available():
        from ._lib import RlviError
        raise RlviError("rlvi_amd.standard needs an MI355X HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def update_weights(losses, tol=1e-3, maxiter=100):
    '''Optimize Bernoulli probabilities (reference rlvi.py:8-20).'''
    l = torch.from_numpy(np.ascontiguousarray(losses, dtype=np.float64)).to(_dev())
    w, _ = ops.update_weights_f64(l, tol=tol, maxiter=maxiter)
    return w.cpu().numpy()


def _wls(X, y, w):
    """theta = argmin sum_i w_i (y_i - x_i.theta)^2 via sqrt(w)-scaled rows (rlvi.py:70-71,79-80;
    the reference materialises the n x n diag(sqrt(w)), the product is the same matrix)."""
    sw = torch.sqrt(w)
    sol = torch.linalg.lstsq(sw[:, None] * X, (sw * y)[:, None])
    return sol.solution[:, 0]


def linear_regression(X, y, maxiter=100, tol=1e-3, return_info=False):
    dev = _dev()
    X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
    y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(dev)
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
    if return_info:
        return theta.cpu().numpy(), w.cpu().numpy(), outer
    return theta.cpu().numpy()


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
