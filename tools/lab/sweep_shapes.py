"""Lab: the M-step over a grid of shapes (one process: graph of 100 launches per shape over a rotation of buffers that
exceeds the Infinity Cache where the shape allows) -- to spot shapes the launcher serves badly next to their
neighbours.  Prints us per launch and the algorithmic GB/s."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from rlvi_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
dtypes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "bf16"]
Cs = [10, 14, 32, 50, 64, 100, 101, 128, 160, 200, 256, 365, 512, 768, 1000, 1001, 2000, 4096]
Bs = [4096, 16384, 65536]
side = torch.cuda.Stream()
L = _lib.load()
print("rows " + " ".join(f"{c:>12d}" for c in Cs))
for dt in dtypes:
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    for B in Bs:
        cells = []
        for C in Cs:
            es = 4 if dt == "f32" else 2
            if B * C * es > 280e6:
                cells.append("           -")
                continue
            nbuf = max(2, min(12, int(600e6 // (2 * B * C * es))))
            with torch.cuda.stream(side):
                zs = [torch.randn(B, C, device=dev).to(tdt) for _ in range(nbuf)]
                gs = [torch.empty_like(z) for z in zs]
                lab = torch.randint(0, C, (B,), device=dev)
                idx = torch.randperm(B, device=dev)
                w = torch.rand(B, device=dev)
                r = torch.zeros(B, device=dev)
                ws = ops.Workspace(dev, B, B)
                for i in range(nbuf):
                    ops.mstep_fwd_bwd(zs[i], lab, idx, w, r, grad=gs[i], ws=ws, accumulate=True)
                side.synchronize()
                K = 60
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for i in range(K):
                        ops.mstep_fwd_bwd(zs[i % nbuf], lab, idx, w, r, grad=gs[i % nbuf], ws=ws, accumulate=True)
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    g.replay()
                    e1.record(side)
                    side.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / K)
                us = sorted(ts)[1]
                form = L.rlvi_workspace_last_mstep_form(ws.ptr)
                ops.mstep_reduce(ws=ws)
                gbs = B * (2 * C * es + 24) / us / 1e3
                cells.append(f"{us:6.1f}/{gbs:4.0f}f{form}")
                del zs, gs, g
        print(f"{dt} {B:6d} " + " ".join(f"{c:>12s}" for c in cells), flush=True)
