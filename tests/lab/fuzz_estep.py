"""Lab: random vectors through the three E-steps (deep fp32 -- trajectory solver on ONE workspace per size class, so
every call starts from an unrelated call's guesses --, standard and online fp64), the in-batch E+M and the one-launch
online batch, against the oracle: iteration counts equal, pi to 1e-5 / 1e-7 (fp64: 1e-9).  A stop test that falls within
a relative 1e-4 of tol on the oracle's own trace is reported as "on the edge", not as a disagreement."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from oracle import rlvi_oracle as oracle  # noqa: E402  (lab: the checker)
from rlvi_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def rel_pi(a, b):
    big = b >= 1e-6 * b.max()
    r = np.abs(a[big] - b[big]) / np.maximum(b[big], 1e-30)
    small = np.abs(a[~big] - b[~big]).max() if (~big).any() else 0.0
    return (r.max() if big.any() else 0.0), small


def vector(N):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        r = rng.exponential(float(rng.choice([0.05, 1.0, 5.0])), N)
    elif kind == 1:                                           # bimodal, random mixture and shift
        r = rng.exponential(0.05, N)
        bad = rng.random(N) >= rng.uniform(0.1, 0.95)
        r[bad] += rng.uniform(2.0, 30.0) + rng.standard_normal(int(bad.sum()))
    elif kind == 2:
        r = np.abs(rng.standard_cauchy(N))                    # heavy tail: exp underflows for many
    elif kind == 3:
        r = rng.exponential(1.0, N)
        r[rng.random(N) < 0.1] = 0.0                          # exact zeros (unvisited slots)
    elif kind == 4:
        r = np.full(N, rng.uniform(0.0, 3.0))                 # all equal
    elif kind == 5:
        r = rng.gamma(rng.uniform(0.3, 5.0), rng.uniform(0.1, 3.0), N) + rng.uniform(0, 100)     # a large common offset
    else:
        r = rng.uniform(0, rng.choice([0.01, 1.0, 50.0]), N)
    return np.abs(r).astype(np.float32), kind


bad = edge = 0
wss = {}
for c in range(cases):
    which = int(rng.integers(0, 5))
    if which <= 1:                                            # deep fp32 E-step
        N = int(np.exp(rng.uniform(0, np.log(300000))))
        r, kind = vector(N)
        tol = float(rng.choice([1e-3, 1e-3, 1e-4, 1e-2]))
        maxiter = int(rng.choice([40, 40, 40, 5, 1, 64, 100]))
        w0 = np.ones(N, np.float32) if rng.random() < 0.5 else rng.random(N).astype(np.float32)
        cls = int(np.log2(max(N, 1))) // 3
        ws = wss.setdefault(cls, ops.Workspace(dev, 300000, 0))
        r_o, w_o = r.copy(), w0.copy()
        it_o, err_o, _ = oracle.update_sample_weights(r_o, w_o, tol=tol, maxiter=maxiter, trace=True)
        rt, wt = torch.from_numpy(r.copy()).to(dev), torch.from_numpy(w0.copy()).to(dev)
        iters = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.estep_deep(rt, wt, tol=tol, maxiter=maxiter, iters=iters, ws=ws)
        torch.cuda.synchronize()
        it = int(iters)
        rel, small = rel_pi(wt.cpu().numpy(), w_o)
        near = bool(len(err_o) and np.min(np.abs(err_o / tol - 1.0)) < 1e-4)
        ok = it == it_o and rel <= 1e-5 and small <= 1e-7 and np.allclose(rt.cpu().numpy(), r_o, rtol=1e-6, atol=1e-6)
        st = ws.status()
        if st:
            ws.clear_status()
        if not ok or st:
            if near and it != it_o:
                edge += 1
            else:
                bad += 1
            print(f"case {c}: deep N={N} kind={kind} tol={tol} maxiter={maxiter}: iters {it} / {it_o}, pi rel {rel:.2e} small {small:.2e}, status {st}" + (" (stop test on the edge)" if near else ""))
    elif which == 2:                                          # fp64 E-steps
        n = int(np.exp(rng.uniform(0, np.log(100000))))
        r, kind = vector(n)
        l = r.astype(np.float64)
        online = rng.random() < 0.5
        tol = float(rng.choice([1e-3, 1e-3, 1e-5]))
        maxiter = int(rng.choice([100, 100, 7, 1]))
        if online:
            w_o, it_o = oracle.update_weights_rlvi(l, tol=tol, maxiter=maxiter, trace=True)
        else:
            w_o, it_o, _ = oracle.update_weights(l, tol=tol, maxiter=maxiter, trace=True)
        w, iters = ops.update_weights_f64(torch.from_numpy(l).to(dev), tol=tol, maxiter=maxiter, online=online)
        torch.cuda.synchronize()
        e = float(np.max(np.abs(w.cpu().numpy() - w_o)))
        st = ops.workspace(dev).status()
        if st:
            ops.workspace(dev).clear_status()
        if int(iters) != it_o or e > 1e-9 * max(1.0, float(np.max(np.abs(w_o)))) or st:
            bad += 1
            print(f"case {c}: f64 {'online' if online else 'standard'} n={n} kind={kind} tol={tol} maxiter={maxiter}: iters {int(iters)} / {it_o}, abs {e:.2e}, status {st}")
    elif which == 3:                                          # in-batch E+M
        B = int(np.exp(rng.uniform(np.log(2), np.log(70000))))
        C = int(rng.integers(2, 200))
        if B * C > 8_000_000:
            B = 8_000_000 // C
        d = synth.mstep_inputs(B, C, seed=9000 + c, clean_frac=float(rng.uniform(0.2, 0.9)), shift=float(rng.uniform(3, 15)))
        pi0 = np.ones(B, np.float32)
        pit = torch.from_numpy(pi0.copy()).to(dev)
        ws = wss.setdefault("fused", ops.Workspace(dev, 70000, 70000))
        out, grad, rows, iters = ops.fused_em(torch.from_numpy(d["logits"]).to(dev), torch.from_numpy(d["labels"]).to(dev), pit, ws=ws)
        torch.cuda.synchronize()
        loss, _ = oracle.nll_rows(d["logits"], d["labels"])
        l2, w2 = loss.copy(), pi0.copy()
        it_o, err_o, _ = oracle.update_sample_weights(l2, w2, trace=True)
        ref = oracle.mstep(d["logits"], d["labels"], np.arange(B), w2, np.zeros(B, np.float32))
        rel, small = rel_pi(pit.cpu().numpy(), w2)
        diff = grad.cpu().numpy().astype(np.float64) - ref["grad"]
        grel = np.sqrt((diff ** 2).sum()) / max(np.sqrt((ref["grad"].astype(np.float64) ** 2).sum()), 1e-30)
        lrel = abs(float(out[0]) - float(ref["loss"])) / max(abs(float(ref["loss"])), 1e-30)
        near = bool(len(err_o) and np.min(np.abs(err_o / 1e-3 - 1.0)) < 1e-4)
        st = ws.status()
        if st:
            ws.clear_status()
        # (a batch of a few clean rows: gradient entries are p - onehot at p = 0.9999, the loss log(1 + 1e-4) -- good to
        #  one fp32 ulp of 1 on both sides, which is more than 1e-5 of such a number)
        gbad = grel > 1e-5 and np.abs(diff).max() > 1.2e-7
        lbad = lrel > 1e-5 and abs(float(out[0]) - float(ref["loss"])) > 2e-7
        if int(iters) != it_o or rel > 1e-5 or small > 1e-7 or gbad or lbad or st:
            if near and int(iters) != it_o:
                edge += 1
            else:
                bad += 1
            print(f"case {c}: fused B={B} C={C}: iters {int(iters)} / {it_o}, pi rel {rel:.2e}, grad rel {grel:.2e}, loss rel {lrel:.2e}, status {st}" + (" (stop test on the edge)" if near else ""))
    else:                                                     # one online mini-batch in one launch
        n, dd = int(np.exp(rng.uniform(0, np.log(4096)))), int(np.exp(rng.uniform(0, np.log(700))))
        Xl, wl, b = synth.logistic_data(n, dd, seed=3000 + c)
        Xl = Xl * float(rng.choice([1.0, 1.0, 5.0]))
        l_o = oracle.logistic_nll(Xl, wl, b)
        w_o, it_o = oracle.update_weights_rlvi(l_o, trace=True)
        losses = torch.empty(n, dtype=torch.float64, device=dev)
        iters = torch.zeros(1, dtype=torch.int32, device=dev)
        w, _, _ = ops.sample_weight_online(torch.from_numpy(Xl).to(dev), torch.from_numpy(wl).to(dev), b, losses=losses, iters=iters)
        torch.cuda.synchronize()
        e = float(np.max(np.abs(w.cpu().numpy() - w_o)))
        el = float(np.max(np.abs(losses.cpu().numpy() - l_o) / np.maximum(np.abs(l_o), 1e-12)))
        if int(iters) != it_o or e > 1e-9 or el > 1e-9:
            bad += 1
            print(f"case {c}: online batch n={n} d={dd}: iters {int(iters)} / {it_o}, weights abs {e:.2e}, losses rel {el:.2e}")
print(f"{cases} cases: {bad} disagreements, {edge} stop tests on the edge")
