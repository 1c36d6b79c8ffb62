import sys, numpy as np, torch
sys.path.insert(0, '.')
from rlvi_amd import ops, synth
from oracle import rlvi_oracle as O
dev = torch.device('cuda:0')
for N in (30000, 5000, 300):
    for case in ("inf_some", "huge", "neg", "nan_one", "all_equal_big", "tiny_spread"):
        r = synth.residual_vector("bimodal", N, seed=3)
        if case == "inf_some": r[::7] = np.inf
        if case == "huge": r[::5] = 1e30
        if case == "neg": r = r - 50.0
        if case == "nan_one": r[11] = np.nan
        if case == "all_equal_big": r[:] = 1e20
        if case == "tiny_spread": r = (1.0 + 1e-7 * np.arange(N)).astype(np.float32)
        ws = ops.Workspace(dev, N, 0)
        rt, wt = torch.from_numpy(r.copy()).to(dev), torch.ones(N, device=dev)
        it = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.estep_deep(rt, wt, iters=it, ws=ws)
        torch.cuda.synchronize()
        rr, ww = r.copy(), np.ones(N, np.float32)
        with np.errstate(all="ignore"):
            ito = O.update_sample_weights(rr, ww)
        w = wt.cpu().numpy()
        fin = np.isfinite(ww)
        rel = np.max(np.abs(w[fin] - ww[fin]) / np.maximum(ww[fin], 1e-30)) if fin.any() and ww[fin].max() > 0 else float('nan')
        print(N, case, "it gpu/oracle", int(it), ito, "status", ws.status(), "nan gpu/oracle", int(np.isnan(w).sum()), int(np.isnan(ww).sum()), "rel %.2e" % rel, flush=True)
