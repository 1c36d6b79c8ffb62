"""Lab: what do the M-step's random 4-byte gather (pi[idx]) and scatter (residuals[idx] = nll) cost at
65 536 x 100?  The launch with both, without the gather (weights = NULL), without the scatter (residuals = NULL),
without either, and with in-order indices; graph of 200 launches over the rotation of bench.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from rlvi_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B, C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 100
d0, labels, idx, logits, grads, weights, residuals = bench.make_inputs(torch, dev, B, C, B, 0)
seq = torch.arange(B, device=dev, dtype=torch.int64)
L = _lib.load()
side = torch.cuda.Stream()
K = 200
with torch.cuda.stream(side):
    ws = ops.Workspace(dev, B, B)

    def leg(i, ix, w, r):
        k = i % bench.ROTATE
        rc = L.rlvi_mstep_fwd_bwd_f32(ops._ptr(logits[k]), C, ops._ptr(labels), ops._ptr(ix), ops._ptr(w), ops._ptr(r),
                                      B, B, C, 1.0 / B, ops._ptr(grads[k]), C, None, ws.ptr, ops._stream_ptr())
        _lib.check(rc, "mstep")

    ones = torch.ones_like(weights)
    zero_ix = torch.zeros_like(idx)
    for name, ix, w, r in (("gather + scatter, permuted", idx, weights, residuals),
                           ("in-order gather of all-ones weights, no scatter", None, ones, None),
                           ("in-order gather, no scatter", None, weights, None),
                           ("every row gathers weights[0], no scatter", zero_ix, weights, None),
                           ("permuted gather, no scatter", idx, weights, None),
                           ("no gather (pi = 1), in-order scatter", None, None, residuals),
                           ("neither", None, None, None),
                           ("in-order indices", seq, weights, residuals),
                           ("no index vector (idx = NULL)", None, weights, residuals),
                           ("gather + scatter, permuted (again)", idx, weights, residuals)):
        for i in range(24):
            leg(i, ix, w, r)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                leg(i, ix, w, r)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            side.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / K)
        ts.sort()
        print(f"{name:40s} {ts[2]:7.2f} us/launch  (min {ts[0]:.2f})  status {ws.status()}")
        ops.mstep_reduce(ws=ws)
