import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rlvi_amd import ops, synth
from oracle import rlvi_oracle as O
dev = torch.device("cuda:0")
B, C = 65536, 100
d = synth.mstep_inputs(B, C, N=B, seed=synth.BENCH_SEED)
z = torch.from_numpy(d["logits"]).to(dev); lab = torch.from_numpy(d["labels"]).to(dev); idx = torch.from_numpy(d["idx"]).to(dev)
w = torch.from_numpy(d["weights"]).to(dev); res = torch.zeros(B, device=dev)
out, grad = ops.mstep_fwd_bwd(z, lab, idx, w, res)
torch.cuda.synchronize()
r0 = np.zeros(B, np.float32)
ref = O.mstep(d["logits"], d["labels"], d["idx"], d["weights"], r0)
g = grad.cpu().numpy()
print("grad equal bits:", np.array_equal(g, ref["grad"]), "max abs diff", np.abs(g.astype(np.float64) - ref["grad"]).max(), ref["grad"].dtype)
print("res equal:", np.array_equal(res.cpu().numpy(), r0), np.abs(res.cpu().numpy() - r0).max())
print("sample", g[0, :4], ref["grad"][0, :4])
