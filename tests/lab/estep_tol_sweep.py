import sys, numpy as np, torch
sys.path.insert(0, '.')
from rlvi_amd import ops, synth
from oracle import rlvi_oracle as O
dev = torch.device('cuda:0')
for N in (65536, 12000):
    for kind in ("exp", "zeros10", "bimodal"):
        for tol, maxiter in ((1e-5, 64), (1e-6, 64), (1e-2, 5), (1e-3, 3), (0.0, 20), (1e-4, 40)):
            r = synth.residual_vector(kind, N, seed=2)
            ws = ops.Workspace(dev, N, 0)
            for rep in range(2):
                rt, wt = torch.from_numpy(r.copy()).to(dev), torch.ones(N, device=dev)
                it = torch.zeros(1, dtype=torch.int32, device=dev)
                ops.estep_deep(rt, wt, tol=tol, maxiter=maxiter, iters=it, ws=ws)
            torch.cuda.synchronize()
            rr, ww = r.copy(), np.ones(N, np.float32)
            ito, err, _ = O.update_sample_weights(rr, ww, tol=tol, maxiter=maxiter, trace=True)
            w = wt.cpu().numpy()
            big = ww >= 1e-6 * ww.max()
            rel = np.max(np.abs(w[big] - ww[big]) / ww[big])
            tie = np.min(np.abs(err - tol)) < 1e-5 * max(tol, 1e-30)
            print(N, kind, tol, maxiter, "it", int(it), ito, "rel %.1e" % rel, "status", ws.status(), "tie" if tie else "", flush=True)
