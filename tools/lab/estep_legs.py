"""Lab: the E-step legs of bench.py alone (hipGraph of 100 calls each): the same vector every call, drifting vectors,
no guess at all (workspace option cold_start), at N = 65 536."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rlvi_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "bimodal"
base = synth.residual_vector(kind, N, seed=5)
rng = np.random.default_rng(11)
dr = [torch.from_numpy((base * np.float32(1.02 ** k) + np.float32(0.01) * rng.random(N).astype(np.float32))
                       .astype(np.float32)).to(dev) for k in range(5)]
walk = [0, 1, 2, 3, 4, 3, 2, 1]
w = torch.ones(N, device=dev)
it = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def timed(fn, K=100):
    with torch.cuda.stream(side):
        for i in range(8):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                fn(i)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(ts)[2]


with torch.cuda.stream(side):
    ws_w, ws_d, ws_c = ops.Workspace(dev, N, 0), ops.Workspace(dev, N, 0), ops.Workspace(dev, N, 0)
    ws_c.set_option("cold_start", 1)
print(f"N={N} {kind}: same vector {timed(lambda i: ops.estep_deep(dr[0], w, iters=it, ws=ws_w)):.2f} us  "
      f"drifting {timed(lambda i: ops.estep_deep(dr[walk[i % 8]], w, iters=it, ws=ws_d)):.2f} us  "
      f"cold {timed(lambda i: ops.estep_deep(dr[walk[i % 8]], w, iters=it, ws=ws_c)):.2f} us  iters {int(it)}  "
      f"status {ws_w.status() | ws_d.status() | ws_c.status()}", flush=True)
