"""End-to-end latency of the numpy-in / numpy-out paths (cfg1, cfg2 shapes), host round trips included."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rlvi_amd import online, standard, synth  # noqa: E402

X, y = synth.linreg_data(size=1000, d=20, eps=0.3, nu=2.5, seed=0)
standard.linear_regression(X, y)                       # warm-up (library load, workspace)
t0 = time.perf_counter()
for _ in range(20):
    theta = standard.linear_regression(X, y)
dt = (time.perf_counter() - t0) / 20
print(f"linear_regression n=1000 d=20: {dt * 1e3:.2f} ms per call, |theta-1|_max = {np.abs(theta - 1).max():.4f}")
rng = np.random.default_rng(0)
for n in (40, 256, 1000, 100000):
    l = rng.exponential(1.0, n)
    standard.update_weights(l)
    t0 = time.perf_counter()
    for _ in range(50):
        standard.update_weights(l)
    print(f"update_weights n={n}: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us per call")
    online.update_weights_rlvi(l)
    t0 = time.perf_counter()
    for _ in range(50):
        online.update_weights_rlvi(l)
    print(f"update_weights_rlvi (online) n={n}: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us per call")
