"""E-step latency from a COLD workspace (no warm-start trajectory) vs warm, and rounds taken."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rlvi_amd import ops, synth  # noqa: E402

SCRATCH_OFF = ops.debug_scratch_offset()
dev = torch.device("cuda:0")
for N in [int(a) for a in sys.argv[1:]]:
    for kind in ("bimodal", "ce", "heavy"):
        r = synth.residual_vector(kind, N, seed=1)
        wss = [ops.Workspace(dev, N, 0) for _ in range(12)]
        iters = torch.zeros(1, dtype=torch.int32, device=dev)
        pairs = [(torch.from_numpy(r.copy()).to(dev), torch.ones(N, device=dev)) for _ in range(12)]
        ops.estep_deep(*pairs[0], iters=iters, ws=wss[0])          # code pages in
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(1, 12):
            ops.estep_deep(*pairs[k], iters=iters, ws=wss[k])      # every call on a fresh workspace
        e1.record()
        torch.cuda.synchronize()
        cold = e0.elapsed_time(e1) * 1000 / 11
        rounds = ""
        if os.environ.get("RLVI_TJ_DEBUG"):
            raw = wss[11].buf[SCRATCH_OFF:SCRATCH_OFF + 100 * 8].cpu().numpy().view(np.uint64)
            rounds = " rounds(Ke,it,delta): " + str([(int(x >> 40), int((x >> 32) & 0xFF), round(float(
                np.array([x & 0xFFFFFFFF], np.uint32).view(np.float32)[0]), 6)) for x in raw[64:76] if x])
        print(N, kind, "it", int(iters), "cold us/call %.1f" % cold, rounds, flush=True)
