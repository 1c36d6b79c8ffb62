"""Lab: where the one-launch linear_regression spends its time (n = 1000, d = 20): the launch with maxiter = 0 (one
weighted solve + residuals), with one inner iteration per E-step, and the full estimator, each as a hipGraph of 50."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rlvi_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 20)
X, y = synth.linreg_data(n, d, seed=0)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
th = torch.empty(d, dtype=torch.float64, device=dev)
w = torch.empty(n, dtype=torch.float64, device=dev)
info = torch.zeros(4, dtype=torch.int32, device=dev)
ws = ops.Workspace(dev, n, 0)
side = torch.cuda.Stream()


def timed(kw, K=50):
    with torch.cuda.stream(side):
        for _ in range(3):
            ops.linear_regression(Xd, yd, theta=th, weights=w, info=info, ws=ws, **kw)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(K):
                ops.linear_regression(Xd, yd, theta=th, weights=w, info=info, ws=ws, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            e0.record(side)
            g.replay()
            e1.record(side)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(ts)[1], info.cpu().numpy().tolist()


for name, kw in (("maxiter=0 (1 solve + residuals)", dict(maxiter=0)),
                 ("maxiter=1, 1 inner iteration", dict(maxiter=1, estep_maxiter=1)),
                 ("maxiter=2, 1 inner iteration each", dict(maxiter=2, estep_maxiter=1)),
                 ("maxiter=1, 50 inner iterations", dict(maxiter=1, estep_maxiter=50, estep_tol=0.0)),
                 ("maxiter=1, 100 inner iterations", dict(maxiter=1, estep_maxiter=100, estep_tol=0.0)),
                 ("full", dict())):
    us, inf = timed(kw)
    print(f"{name:40s} {us:8.1f} us   info {inf}", flush=True)

if os.environ.get("RLVI_LIB_PATH"):
    # a -DRLVI_STAMPS=1 build: the phase stamps of one eager call (100 MHz ticks -> us, relative to the first)
    ops.linear_regression(Xd, yd, theta=th, weights=w, info=info, ws=ws)
    torch.cuda.synchronize()
    off = ops.debug_scratch_offset()
    raw = ws.buf[off:off + 120 * 8].cpu().numpy().view(np.uint64)
    names = ["outer start", "gram", "block sums", "solved", "residuals", "exp", "E-step"]
    t0 = int(raw[0])
    print("stamps (us after the first):")
    print(" ".join(f"{(int(x) - t0) / 100.0:.2f}" for x in raw[:40] if x))
