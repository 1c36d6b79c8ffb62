"""Seeded synthetic inputs for the RLVI hot path (numpy only).

Shared by bench.py, the parity tests and oracle/make_golden.py so that a
fixture can carry a recipe name + seed instead of a large array.  Recipes follow
SURVEY.md section 8(d).
"""
import numpy as np

BENCH_SEED = 20240131


def mstep_inputs(B, C, N=None, seed=BENCH_SEED, clean_frac=0.55, shift=12.0,
                 zero_frac=0.0):
    """logits ~ 3*N(0,1); `clean_frac` of the rows get +shift on the label logit
    (bimodal NLL); idx = first B of a permutation of N; lagged pi ~ U(0,1)."""
    rng = np.random.default_rng(seed)
    N = B if N is None else N
    logits = (3.0 * rng.standard_normal((B, C))).astype(np.float32)
    labels = rng.integers(0, C, B).astype(np.int64)
    clean = rng.random(B) < clean_frac
    logits[np.nonzero(clean)[0], labels[clean]] += np.float32(shift)
    idx = rng.permutation(N)[:B].astype(np.int64)
    weights = rng.random(N).astype(np.float32)
    if zero_frac > 0:
        weights[rng.random(N) < zero_frac] = 0.0
    return dict(logits=logits, labels=labels, idx=idx, weights=weights,
                residuals=np.zeros(N, np.float32))


def residual_vector(kind, N, seed=0):
    """Per-sample NLL vectors that exercise the E-step (golden set G1)."""
    rng = np.random.default_rng(seed)
    if kind == "equal":
        r = np.full(N, 0.7, np.float32)
    elif kind == "exp":
        r = rng.exponential(1.0, N).astype(np.float32)
    elif kind == "bimodal":
        r = rng.exponential(0.05, N).astype(np.float32)
        bad = rng.random(N) >= 0.55
        r[bad] += (12.0 + rng.standard_normal(int(bad.sum()))).astype(np.float32)
    elif kind == "heavy":
        r = rng.exponential(1.0, N).astype(np.float32)
        r[::7] += np.float32(95.0)              # exp(-l) underflows to 0 in fp32
    elif kind == "zeros10":
        r = (0.3 + rng.exponential(1.0, N)).astype(np.float32)
        r[rng.random(N) < 0.10] = 0.0           # Food-101 unvisited-slot quirk
    elif kind == "ce":
        d = mstep_inputs(N, 10, seed=seed)
        z = d["logits"].astype(np.float64)
        m = z.max(1, keepdims=True)
        lse = m[:, 0] + np.log(np.exp(z - m).sum(1))
        r = (lse - z[np.arange(N), d["labels"]]).astype(np.float32)
    else:
        raise ValueError(kind)
    return r


def linreg_data(size=1000, d=20, eps=0.3, nu=2.5, seed=0):
    """cfg1 inputs: the recipe of standard-learning/main.py:69-85 at size x d."""
    rng = np.random.default_rng(seed)
    X = -5 + 10 * rng.random(size=(size, d))
    n2 = rng.binomial(n=size, p=eps)
    n1 = size - n2
    theta = np.ones(d)
    y1 = X[:n1] @ theta + 0.25 * rng.normal(size=n1)
    u = rng.chisquare(df=nu, size=n2) / nu
    v = rng.normal(size=n2)
    y2 = X[n1:] @ theta + v / np.sqrt(u)
    return X, np.concatenate([y1, y2])


def logistic_data(B=256, d=60, seed=0):
    """cfg2 stand-in (the HAR .mat is not shipped): X ~ N(0,1), w ~ N(0,1)/sqrt(d)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    b = 0.1
    return X, w, b


def heavy_tail_cloud(size=200, eps=0.2, nu=1.5, seed=0, corr=0.8):
    """2-D sample for the mean / pca / covariance estimators: a correlated Gaussian cloud with an
    eps-fraction of multivariate-t outliers (the recipe of standard-learning/main.py:44-67,
    :124-167 with a fixed seed)."""
    rng = np.random.default_rng(seed)
    n2 = rng.binomial(n=size, p=eps)
    n1 = size - n2
    root = np.linalg.cholesky(np.array([[1.0, corr], [corr, 1.0]]))
    s1 = root @ rng.normal(size=(2, n1))
    u = rng.chisquare(df=nu, size=n2) / nu
    s2 = (root @ rng.normal(size=(2, n2))) / np.sqrt(u[None, :])
    return np.hstack([s1, s2]).T
