"""Lab: the in-batch E+M over a grid of shapes, as the launcher serves it and as three launches (RLVI_FUSED_EM=0) --
to see where the one-launch forms end and whether any shape is served worse than the composition."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from rlvi_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
side = torch.cuda.Stream()
Cs = [10, 14, 16, 32, 64, 100, 101, 128, 200, 1000]
Bs = [256, 1024, 4096, 16384, 65536]
print("rows  " + " ".join(f"{c:>13d}" for c in Cs))
for B in Bs:
    cells = []
    for C in Cs:
        if B * C * 4 > 120e6:
            cells.append("            -")
            continue
        d = synth.mstep_inputs(B, C, seed=1)
        nbuf = max(2, min(8, int(400e6 // (2 * B * C * 4))))
        res = []
        with torch.cuda.stream(side):
            zs = [torch.from_numpy(d["logits"]).to(dev) + 0.01 * i for i in range(nbuf)]
            gs = [torch.empty_like(z) for z in zs]
            lab = torch.from_numpy(d["labels"]).to(dev)
            pi = torch.ones(B, device=dev)
            rows = torch.empty(B, device=dev)
            out = torch.empty(4, device=dev)
            it = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = ops.Workspace(dev, B, B)
            for mode in (1, 0):
                _lib.check(L.rlvi_tune_set(b"RLVI_FUSED_EM", mode), "tune")
                for i in range(nbuf):
                    ops.fused_em(zs[i], lab, pi, ws=ws, out=out, grad=gs[i], rows=rows, iters=it)
                side.synchronize()
                K = 40
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for i in range(K):
                        ops.fused_em(zs[i % nbuf], lab, pi, ws=ws, out=out, grad=gs[i % nbuf], rows=rows, iters=it)
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    g.replay()
                    e1.record(side)
                    side.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / K)
                res.append(sorted(ts)[1])
                del g
            L.rlvi_tune_unset(b"RLVI_FUSED_EM")          # (returns 1: there was a set value)
            st = ws.status()
        cells.append(f"{res[0]:5.1f}/{res[1]:5.1f}" + ("!" if st else " "))
        del zs, gs
    print(f"{B:6d} " + " ".join(f"{c:>13s}" for c in cells), flush=True)
print("(as served / as three launches, us per call; ! = a status flag was raised)")
