"""Lab: the one-launch linear_regression over a grid of (n, d): us per call in a graph of 30, outer / inner iterations."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from rlvi_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
ds = [2, 10, 12, 16, 20, 23, 24, 31]
print("n     " + " ".join(f"{d:>16d}" for d in ds))
for n in (40, 200, 1000, 1024, 1025, 2000, 4096):
    cells = []
    for d in ds:
        if n <= d + 8:
            cells.append("               -")
            continue
        X, y = synth.linreg_data(n, d, seed=n + d)
        with torch.cuda.stream(side):
            Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
            th = torch.empty(d, dtype=torch.float64, device=dev)
            w = torch.empty(n, dtype=torch.float64, device=dev)
            info = torch.zeros(4, dtype=torch.int32, device=dev)
            ws = ops.Workspace(dev, 4096, 0)
            for _ in range(3):
                ops.linear_regression(Xd, yd, theta=th, weights=w, info=info, ws=ws)
            side.synchronize()
            K = 30
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(K):
                    ops.linear_regression(Xd, yd, theta=th, weights=w, info=info, ws=ws)
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                g.replay()
                e1.record(side)
                side.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / K)
            i = info.cpu().numpy()
        cells.append(f"{sorted(ts)[1]:7.1f} {i[0]:2d}/{i[2]:4d}" + ("F" if i[3] else " "))
    print(f"{n:5d} " + " ".join(f"{c:>16s}" for c in cells), flush=True)
print("(us per call, outer / inner iterations in all; F = the launch asked for the general path)")
