"""Runs the round's new kernels a few hundred times each (for rocprofv3 --kernel-trace --stats): the one-launch
linear_regression at cfg1's shape and at the reference's own, the one-launch online batch at cfg2's shape, the
word-wise bf16 M-step at 65 536 x 101, the flat copy."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rlvi_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
ws = ops.Workspace(dev, 65536, 65536)
for n, d in ((1000, 20), (40, 10)):
    X, y = synth.linreg_data(n, d, seed=0)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    th = torch.empty(d, dtype=torch.float64, device=dev)
    w = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(200):
        ops.linear_regression(Xd, yd, theta=th, weights=w, info=info, ws=ws)
Xl, wl, b = synth.logistic_data(256, 60)
Xld, wld = torch.from_numpy(Xl).to(dev), torch.from_numpy(wl).to(dev)
sw = torch.empty(256, dtype=torch.float64, device=dev)
for _ in range(200):
    ops.sample_weight_online(Xld, wld, b, out=sw)
B, C = 65536, 101
dd = synth.mstep_inputs(B, C, seed=1)
zs = [torch.from_numpy(dd["logits"]).to(dev).to(torch.bfloat16) + k for k in range(6)]
gs = [torch.empty_like(z) for z in zs]
lab, idx = torch.from_numpy(dd["labels"]).to(dev), torch.from_numpy(dd["idx"]).to(dev)
wt = torch.from_numpy(dd["weights"]).to(dev)
res = torch.zeros(B, device=dev)
for i in range(120):
    ops.mstep_fwd_bwd(zs[i % 6], lab, idx, wt, res, grad=gs[i % 6], ws=ws, accumulate=True)
ops.mstep_reduce(ws=ws)
a = [torch.empty(65536 * 100, device=dev) for _ in range(12)]
for i in range(120):
    ops.stream_copy(a[(i + 6) % 12], a[i % 12])
torch.cuda.synchronize()
print("ok")
# precision@k (topk.hip) at the bench shape and at cfg3's
for Bk, Ck in ((65536, 100), (4096, 10)):
    dk = synth.mstep_inputs(Bk, Ck, seed=2)
    zk, yk = torch.from_numpy(dk["logits"]).to(dev), torch.from_numpy(dk["labels"]).to(dev)
    hk = torch.empty(2, dtype=torch.int32, device=dev)
    for _ in range(100):
        ops.topk_hits(zk, yk, (1, min(5, Ck)), out=hk)
torch.cuda.synchronize()
print("topk ok")
