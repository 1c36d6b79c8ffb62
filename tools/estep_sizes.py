"""E-step latency by population size (eager launches, warm workspace, rotating inputs).

usage: python tools/estep_sizes.py N [N ...]     (RLVI_TJ_DEBUG=1 adds the in-kernel stamps)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rlvi_amd import ops, synth  # noqa: E402

SCRATCH_OFF = ops.debug_scratch_offset()

dev = torch.device("cuda:0")
for N in [int(a) for a in sys.argv[1:]]:
    ws = ops.Workspace(dev, N, 0)
    rs = [torch.from_numpy(synth.residual_vector("bimodal", N, seed=s)).to(dev) for s in range(4)]
    iters = torch.zeros(1, dtype=torch.int32, device=dev)

    def pair(k):
        return rs[k % 4].clone(), torch.ones(N, device=dev)

    for k in range(3):
        rt, wt = pair(k)
        ops.estep_deep(rt, wt, iters=iters, ws=ws)
    torch.cuda.synchronize()
    bufs = [pair(k) for k in range(20)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rt, wt in bufs:
        ops.estep_deep(rt, wt, iters=iters, ws=ws)
    e1.record()
    torch.cuda.synchronize()
    print(N, "it", int(iters), "us/call %.1f" % (e0.elapsed_time(e1) * 1000 / 20), "status", ws.status(),
          flush=True)
    if os.environ.get("RLVI_TJ_DEBUG"):
        raw = ws.buf[SCRATCH_OFF:SCRATCH_OFF + 700 * 8].cpu().numpy().view(np.uint64)
        n = int(raw[63])
        st = raw[:n].astype(np.int64)
        print("   stamps us:", [round(float(x - st[0]) / 100.0, 2) for x in st][:24])
        rd = raw[64:88]
        print("   rounds (Ke, it, delta):",
              [(int(x >> 40), int((x >> 32) & 0xFF),
                float(np.array([x & 0xFFFFFFFF], np.uint32).view(np.float32)[0])) for x in rd if x][:12])
        nn = raw[130:130 + 24]
        rn = np.array(nn >> 32, np.uint32).view(np.float32).astype(np.float64)
        rnew = np.array(nn & 0xFFFFFFFF, np.uint32).view(np.float32)
        tot = raw[104:128].view(np.float64).reshape(8, 3)
        e = np.exp(-bufs[-1][0].cpu().numpy().astype(np.float64))
        print("   scale", np.array([raw[103] & 0xFFFFFFFF], np.uint32).view(np.float32))
        for k in range(4):
            r = rn[k]
            print("   node", k, "rn", r, "rnew", rnew[k], "kernel S,P,Q", tot[k], "numpy S,P,Q",
                  np.sum(r * e / (1 + r * e)), np.sum(e / (1 + r * e) ** 2), np.sum(e * e / (1 + r * e) ** 3))
        for xs in range(3):
            nn = raw[400 + xs * 64:400 + xs * 64 + 24]
            a = np.array(nn >> 32, np.uint32).view(np.float32).astype(np.float64)
            b = np.array(nn & 0xFFFFFFFF, np.uint32).view(np.float32).astype(np.float64)
            print("   round", xs, "rel change per node:", np.array2string(np.abs(b - a) / a, precision=2, max_line_width=250))
