import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rlvi_amd import ops, synth
from oracle import rlvi_oracle as O
dev = torch.device("cuda:0")
for N in (4096, 65536):
    for use_trace in (False, True):
        r0 = synth.residual_vector("exp", N, 5)
        ws = ops.Workspace(dev, N, N)
        for rep in range(3):
            r = torch.from_numpy(r0.copy() + np.float32(0.01 * rep)).to(dev)
            w = torch.ones(N, device=dev)
            it = torch.zeros(1, dtype=torch.int32, device=dev)
            tr = torch.zeros(80, device=dev) if use_trace else None
            ops.estep_deep(r, w, iters=it, trace=tr, ws=ws)
            ro, wo = r0.copy() + np.float32(0.01 * rep), np.ones(N, np.float32)
            ito = O.update_sample_weights(ro, wo)
            rel = np.abs(w.cpu().numpy() - wo).max()
            print(N, use_trace, rep, "it", int(it), ito, "min-shift equal", np.array_equal(r.cpu().numpy(), ro), "pi maxabs", rel, "status", ws.status())
N = 4096
r0 = synth.residual_vector("exp", N, 5)
ws = ops.Workspace(dev, N, N)
r = torch.from_numpy(r0.copy()).to(dev); w = torch.ones(N, device=dev)
ops.estep_deep(r, w, ws=ws)
g = r.cpu().numpy()
print("true min", r0.min(), "argmin", r0.argmin(), "gpu - (r0 - min):", (g - (r0 - r0.min()))[:6], "gpu[argmin]", g[r0.argmin()], "gpu min", g.min())
print("implied shift", (r0 - g)[:4])
from rlvi_amd import _lib
_lib.check(_lib.load().rlvi_tune_set(b"RLVI_TJ_DEBUG", 1), "t")
N = 4096
r0 = synth.residual_vector("exp", N, 5)
ws = ops.Workspace(dev, N, N)
r = torch.from_numpy(r0.copy()).to(dev); w = torch.ones(N, device=dev)
ops.estep_deep(r, w, ws=ws)
torch.cuda.synchronize()
off = ops.debug_scratch_offset()
full = ws.buf[off:off + 1000 * 8].cpu().numpy().view(np.uint64)
print("wmin stored (wg0):", np.array([full[898] & 0xFFFFFFFF], np.uint32).view(np.float32), " totals val[4] lanes 0..3:", np.array(full[900:904] & 0xFFFFFFFF, np.uint32).view(np.float32), "nq", int(full[900] >> 32))
print("local min of wg0 slice:", r0[:16].min(), "global", r0.min())
g = r.cpu().numpy()
print("DEBUG RUN: min-shift equal", np.array_equal(g, r0 - r0.min()), "implied shift", (r0 - g)[:4], "res_min dbg[100]:", np.array([full[100] & 0xFFFFFFFF], np.uint32).view(np.float32))
