"""Lab: random shapes and contaminations through the one-launch linear_regression against the oracle (scipy lstsq +
the pinned C E-step): theta, weights, outer-iteration count; where the launch answers "fallback", the numpy mirror's
general path instead.  Prints every disagreement and says whether the case was WELL-POSED -- at least d + 8 clean
samples -- or near-interpolating: there the robust fit gives all but about d samples a zero weight, the weighted system
is numerically rank-deficient, and what comes out is decided by which singular values the solver still counts (the
reference's gelsd: down to eps x the largest; a Gram-matrix solve resolves sqrt(eps)) and by rounding noise over
rounding noise in the variance estimate.  Only well-posed disagreements count."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from oracle import rlvi_oracle as oracle  # noqa: E402  (lab: the checker)
from rlvi_amd import ops, standard, synth  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = fell = soft = 0
worst_t = worst_w = 0.0
for c in range(cases):
    d = int(rng.integers(1, 32))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        n = int(rng.integers(d + 1, d + 8))            # barely determined
    elif kind == 1:
        n = int(rng.integers(1, d + 1))                # under-determined: the launch has to say "fallback"
    else:
        n = int(rng.integers(d + 1, 4097))
    eps = float(rng.choice([0.0, 0.05, 0.3, 0.45]))
    X, y = synth.linreg_data(n, d, eps=eps, nu=float(rng.choice([1.0, 2.5, 10.0])), seed=1000 + c)
    if kind == 4:
        y = y * float(rng.choice([1e-6, 1e6]))        # scale of the targets
    if not ops.linear_regression_check(n, d):
        continue
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    theta, w, info = ops.linear_regression(Xd, yd)
    torch.cuda.synchronize()
    info = info.cpu().numpy()
    if info[3]:
        # the launch refuses (the weighted Gram matrix lost its rank: an interpolating fit gives all but d samples a
        # zero weight): the numpy mirror then takes the general path (minimum norm, as the reference's lstsq)
        fell += 1
        try:
            th_o, w_o, outer_o = oracle.linear_regression(X, y, trace=True)
        except ValueError:
            # (scipy's check_finite: an exactly interpolating fit -> variance 0 -> 0/0 losses; whether rounding leaves
            #  an exact zero there is not ours to reproduce, only that nothing silent comes back)
            try:
                th_m = standard.linear_regression(X, y)
                if not np.allclose(X @ th_m, y, rtol=1e-6, atol=1e-9):
                    print(f"case {c}: n={n} d={d}: the reference raises, the mirror returned a non-interpolating fit")
            except ValueError:
                pass
            ops.workspace(dev).clear_status()
            continue
        try:
            th_m, w_m, outer_m = standard.linear_regression(X, y, return_info=True)
        except ValueError:
            print(f"case {c}: n={n} d={d}: the mirror raises, the reference does not (near-interpolating)")
            soft += 1
            continue
        ops.workspace(dev).clear_status()
        fit_o, fit_m = X @ th_o, X @ th_m
        ef = float(np.max(np.abs(fit_o - fit_m)) / (np.max(np.abs(fit_o)) + 1e-300))
        if outer_m != outer_o or ef > 1e-6:
            posed = n * (1.0 - eps) >= d + 8
            bad += posed
            soft += not posed
            print(f"case {c}: n={n} d={d} eps={eps} {'WELL-POSED' if posed else 'near-interpolating'}: fallback; mirror outer {outer_m} / {outer_o}, fitted values rel {ef:.2e}")
        continue
    try:
        th_o, w_o, outer_o = oracle.linear_regression(X, y, trace=True)    # scipy lstsq + the pinned C E-step
    except ValueError:
        soft += 1
        print(f"case {c}: n={n} d={d}: the reference raises (0/0 losses of an interpolating fit), the launch returned numbers")
        continue
    th, ww = theta.cpu().numpy(), w.cpu().numpy()
    et = float(np.max(np.abs(th - th_o) / (np.abs(th_o) + 1e-10)))
    ew = float(np.max(np.abs(ww - w_o)))
    worst_t, worst_w = max(worst_t, et), max(worst_w, ew)
    # (n within a few rows of d: the fit all but interpolates, the variance estimate and with it every loss is rounding
    #  noise over rounding noise, and a tol-stopped E-step that ends one iteration apart moves the weights by up to tol)
    posed = n * (1.0 - eps) >= d + 8
    if info[0] != outer_o or et > 1e-7 or ew > 1e-6 or not np.isfinite(th).all():
        bad += posed
        soft += not posed
        print(f"case {c}: n={n} d={d} eps={eps} {'WELL-POSED' if posed else 'near-interpolating'}: outer {info[0]} / {outer_o}, theta rel {et:.2e}, weights abs {ew:.2e}")
print(f"{cases} cases: {bad} well-posed disagreements, {soft} near-interpolating ones, {fell} fallbacks, worst theta rel {worst_t:.2e}, worst weights abs {worst_w:.2e}, status {ops.workspace(dev).status()}")
