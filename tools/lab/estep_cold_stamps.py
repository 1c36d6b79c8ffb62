"""Lab (a -DRLVI_STAMPS=1 build, RLVI_TJ_DEBUG=1): in-kernel stamps of ONE cold E-step (workspace option cold_start)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rlvi_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
N = 65536
r = torch.from_numpy(synth.residual_vector("bimodal", N, seed=5)).to(dev)
w = torch.ones(N, device=dev)
it = torch.zeros(1, dtype=torch.int32, device=dev)
ws = ops.Workspace(dev, N, 0)
ws.set_option("cold_start", 1)
for _ in range(3):
    ops.estep_deep(r.clone(), w, iters=it, ws=ws)
torch.cuda.synchronize()
off = ops.debug_scratch_offset()
raw = ws.buf[off:off + 1000 * 8].cpu().numpy().view(np.uint64)
n = int(raw[63])
st = raw[:n].astype(np.int64)
names = ["start", "slice", "sums", "stageA", "gathered", "published", "totals", "recurrence"]
print("iters", int(it), "stamps (us since kernel start):", [round(float(x - st[0]) / 100.0, 2) for x in st])
cs = raw[990:996].astype(np.int64)
print("recurrence wave of round 0 [entry, prepared, chain, trust/tail(+global model), acceptance, out] us:",
      [round(float(v - cs[0]) / 100.0, 2) for v in cs])
rd = raw[64:88]
print("rounds (Ke, it, delta):", [(int(x >> 40), int((x >> 32) & 0xFF), float(np.array([x & 0xFFFFFFFF], np.uint32).view(np.float32)[0])) for x in rd if x][:6])
