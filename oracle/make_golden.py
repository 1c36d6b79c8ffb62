#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--ref /root/reference]

The reference is imported read-only from --ref; nothing of it is copied: the
fixtures hold inputs (or a recipe name + seed from rlvi_amd/synth.py) and the
reference's outputs.  /root/reference does not exist on the GPU box, so tests
only ever read the committed .npz files.

Golden sets:
  G1 estep_deep   update_sample_weights            train_rlvi.py:14-38
  G2 threshold    false_negative_criterion + mask  train_rlvi.py:41-49,:103, main.py:343
  G3 mstep        CE/scatter/gather/mean/backward  train_rlvi.py:85-96
  G4 epoch        whole train_rlvi epochs          train_rlvi.py:52-106
  G5 standard     update_weights, linear/logistic  standard-learning/rlvi.py
  G6 online       update_weights_rlvi, CE          online-learning/main.py:45-58,:84-85
  G7 estimators   mean, pca, covariance            standard-learning/rlvi.py:23-65,:111-144
  G8 small-loss   usdnl.loss_fn, loss_coteaching   train_usdnl.py:16-27, train_coteaching.py:17-35
  G9 top-1 ties   evaluate() on tied maxima        deep-learning/utils.py:48-62
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlvi_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def versions():
    import scipy
    import sklearn
    import torch
    return dict(v_numpy=np.__version__, v_torch=torch.__version__,
                v_scipy=scipy.__version__, v_sklearn=sklearn.__version__)


def save(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    kw.update({k: np.array(v) for k, v in versions().items()})
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


def ref_deep(ref):
    sys.path.insert(0, os.path.join(ref, "deep-learning"))
    import methods  # noqa: F401
    return sys.modules["methods.train_rlvi"]


def run_estep_traced(m, residuals, weights):
    """Call the reference E-step, recording `error` of every iteration."""
    import torch
    errs = []
    orig = torch.norm

    def spy(*a, **k):
        v = orig(*a, **k)
        errs.append(float(v))
        return v
    torch.norm = spy
    try:
        m.update_sample_weights(residuals, weights)
    finally:
        torch.norm = orig
    return np.array(errs, np.float32)


G1_SMALL = [("equal", 1), ("exp", 1), ("exp", 2), ("exp", 5), ("equal", 64), ("exp", 64),
            ("bimodal", 64), ("exp", 1000), ("bimodal", 1000), ("heavy", 1000),
            ("zeros10", 1000), ("ce", 1000), ("exp", 4096), ("bimodal", 4096),
            ("heavy", 4096), ("zeros10", 4096), ("ce", 4096)]
G1_LARGE = [("bimodal", 45000), ("ce", 54000), ("bimodal", 65536), ("exp", 65536),
            ("zeros10", 75750)]
STRIDE = 97


def gen_g1_g2(ref):
    import torch
    m = ref_deep(ref)
    out = {}
    cases = []
    for kind, N in G1_SMALL + G1_LARGE:
        for win in ("ones", "prev"):
            seed = 1000 + N
            r0 = synth.residual_vector(kind, N, seed)
            res = torch.from_numpy(r0.copy())
            w = torch.ones(N)
            if win == "prev":
                # previous epoch's output on a perturbed copy -> exercises the
                # iteration-1 error term that depends on the caller's weights
                rp = torch.from_numpy((r0 * np.float32(1.05)).copy())
                m.update_sample_weights(rp, w)
            w_in = w.numpy().copy()
            errs = run_estep_traced(m, res, w)
            w_out = w.numpy().copy()
            r_out = res.numpy().copy()
            # G2 on the E-step output
            thr_t = m.false_negative_criterion(w.clone())
            thr = np.float32(thr_t.item())
            sw, _ = torch.sort(w, dim=0, descending=True)
            fneg = torch.cumsum(1 - sw, dim=0)
            beta = torch.sum(1 - w) * 0.05
            last_index = int(torch.sum(fneg <= beta).item()) - 1
            wt = w.clone()
            wt[wt < thr_t] = 0
            mask = (wt > thr_t).numpy()
            margin = float((fneg.double() - beta.double()).abs().min() / max(float(beta), 1e-30))
            key = f"{kind}_{N}_{win}"
            cases.append(key)
            large = N > 4096
            sl = slice(None, None, STRIDE) if large else slice(None)
            out[key + "/kind"] = np.array(kind)
            out[key + "/N"] = np.array(N)
            out[key + "/seed"] = np.array(seed)
            out[key + "/win"] = np.array(win)
            if not large:
                out[key + "/res_in"] = r0
                out[key + "/w_in"] = w_in
            out[key + "/w_out"] = w_out[sl]
            out[key + "/res_out"] = r_out[sl]
            out[key + "/w_out_sum"] = np.array(w_out.astype(np.float64).sum())
            out[key + "/w_out_min"] = np.array(w_out.min())
            out[key + "/errs"] = errs
            out[key + "/iters"] = np.array(len(errs))
            out[key + "/thr"] = np.array(thr)
            out[key + "/beta"] = np.array(np.float32(beta.item()))
            out[key + "/last_index"] = np.array(last_index)
            out[key + "/margin"] = np.array(margin)
            out[key + "/kept"] = np.array(int(mask.sum()))
            out[key + "/mask_bits"] = np.packbits(mask)
            out[key + "/w_trunc_sum"] = np.array(wt.numpy().astype(np.float64).sum())
    # hand-made threshold cases: count==1 / many exact-1.0 ties / count==0 wrap
    extra = {
        "ties_ones": np.concatenate([np.ones(500, np.float32),
                                     np.linspace(0.0, 0.9, 500, dtype=np.float32)]),
        "count1": np.concatenate([np.array([1.0], np.float32),
                                  np.full(99, 0.5, np.float32)]),
        "count0_wrap": np.array([0.2, 0.1, 0.3, 0.05], np.float32),
        "all_ones": np.ones(64, np.float32),
    }
    for k, wv in extra.items():
        w = torch.from_numpy(wv.copy())
        thr_t = m.false_negative_criterion(w.clone())
        wt = w.clone()
        wt[wt < thr_t] = 0
        mask = (wt > thr_t).numpy()
        out[f"x_{k}/w"] = wv
        out[f"x_{k}/thr"] = np.array(np.float32(thr_t.item()))
        out[f"x_{k}/w_trunc"] = wt.numpy()
        out[f"x_{k}/mask_bits"] = np.packbits(mask)
        out[f"x_{k}/kept"] = np.array(int(mask.sum()))
    out["cases"] = np.array(cases)
    out["extra"] = np.array(list(extra))
    out["stride"] = np.array(STRIDE)
    save("g1_g2_estep_threshold", **out)


G3_CASES = [(32, 10), (128, 10), (128, 100), (777, 100), (1024, 101), (4096, 10), (5, 5)]


def gen_g3(ref):
    import torch
    from torch.nn import functional as F
    m = ref_deep(ref)
    import utils as dl_utils  # deep-learning/utils.py (accuracy)
    out = {}
    keys = []
    for (B, C) in G3_CASES:
        for dt in ("f32", "bf16"):
            d = synth.mstep_inputs(B, C, N=2 * B, seed=300 + B + C, zero_frac=0.2)
            z = d["logits"]
            if dt == "bf16":
                z = torch.from_numpy(z).to(torch.bfloat16).to(torch.float32).numpy()
            logits = torch.from_numpy(z.copy()).requires_grad_(True)
            labels = torch.from_numpy(d["labels"])
            idx = torch.from_numpy(d["idx"])
            weights = torch.from_numpy(d["weights"].copy())
            residuals = torch.zeros(2 * B)
            # the statements of train_rlvi.py:85-96, minus model/optimizer
            prec, _ = dl_utils.accuracy(logits, labels, topk=(1, 5))
            loss = F.cross_entropy(logits, labels, reduction='none')
            residuals[idx] = loss
            bw = weights[idx]
            lw = (loss * bw).mean()
            lw.backward()
            key = f"B{B}_C{C}_{dt}"
            keys.append(key)
            g = logits.grad.numpy()
            rows = slice(None) if B * C <= 12800 else slice(None, None, 16)
            out[key + "/B"] = np.array(B)
            out[key + "/C"] = np.array(C)
            out[key + "/seed"] = np.array(300 + B + C)
            out[key + "/dtype"] = np.array(dt)
            out[key + "/loss_rows"] = loss.detach().numpy()
            out[key + "/loss"] = np.array(np.float32(lw.item()))
            out[key + "/prec1"] = np.array(np.float32(prec.item()))
            out[key + "/residuals"] = residuals.detach().numpy()
            out[key + "/grad_rows"] = g[rows]
            out[key + "/grad_rowstep"] = np.array(1 if rows == slice(None) else 16)
            out[key + "/grad_fro"] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            out[key + "/grad_colsum"] = g.astype(np.float64).sum(0)
    out["cases"] = np.array(keys)
    save("g3_mstep", **out)


def gen_g4(ref):
    import torch
    m = ref_deep(ref)
    torch.manual_seed(7)
    rng = np.random.default_rng(7)
    N, D, C, B = 256, 16, 10, 64
    X = rng.standard_normal((N, D)).astype(np.float32)
    true_w = rng.standard_normal((D, C)).astype(np.float32)
    y = (X @ true_w).argmax(1).astype(np.int64)
    flip = rng.random(N) < 0.3
    y[flip] = rng.integers(0, C, int(flip.sum()))
    model = torch.nn.Linear(D, C)
    W0 = model.weight.detach().numpy().copy()
    b0 = model.bias.detach().numpy().copy()
    opt = torch.optim.SGD(model.parameters(), lr=0.5, momentum=0.9)
    residuals = torch.zeros(N)
    weights = torch.ones(N)
    threshold = 0
    out = dict(X=X, y=y, W0=W0, b0=b0, N=np.array(N), B=np.array(B), lr=np.array(0.5),
               momentum=np.array(0.9))
    orders = []
    for ep, overfit in enumerate([False, False, True, True]):
        perm = rng.permutation(N)
        orders.append(perm)
        loader = []
        for s in range(0, N, B):
            ix = perm[s:s + B]
            loader.append((torch.from_numpy(X[ix]), torch.from_numpy(y[ix]),
                           torch.from_numpy(ix.astype(np.int64))))
        model.train()
        acc, threshold = m.train_rlvi(loader, model, opt, residuals, weights, overfit, threshold)
        out[f"ep{ep}/residuals"] = residuals.detach().numpy().copy()
        out[f"ep{ep}/weights"] = weights.detach().numpy().copy()
        out[f"ep{ep}/threshold"] = np.array(np.float32(float(threshold)))
        out[f"ep{ep}/train_acc"] = np.array(acc)
        out[f"ep{ep}/W"] = model.weight.detach().numpy().copy()
        out[f"ep{ep}/b"] = model.bias.detach().numpy().copy()
        out[f"ep{ep}/overfit"] = np.array(overfit)
    out["orders"] = np.stack(orders)
    # The same four epochs with model, data, residuals and weights in fp64: how far the reference's
    # own fp32 run drifts from exact arithmetic.  The GPU tests accept a stated multiple of this.
    model64 = torch.nn.Linear(D, C).double()
    with torch.no_grad():
        model64.weight.copy_(torch.from_numpy(W0).double())
        model64.bias.copy_(torch.from_numpy(b0).double())
    opt64 = torch.optim.SGD(model64.parameters(), lr=0.5, momentum=0.9)
    res64 = torch.zeros(N, dtype=torch.float64)
    w64 = torch.ones(N, dtype=torch.float64)
    thr64 = 0
    for ep, overfit in enumerate([False, False, True, True]):
        perm = orders[ep]
        loader = []
        for s in range(0, N, B):
            ix = perm[s:s + B]
            loader.append((torch.from_numpy(X[ix]).double(), torch.from_numpy(y[ix]),
                           torch.from_numpy(ix.astype(np.int64))))
        model64.train()
        _, thr64 = m.train_rlvi(loader, model64, opt64, res64, w64, overfit, thr64)
        out[f"drift/ep{ep}/W"] = np.array(np.abs(model64.weight.detach().numpy() - out[f"ep{ep}/W"]).max())
        out[f"drift/ep{ep}/residuals"] = np.array(np.abs(res64.detach().numpy() - out[f"ep{ep}/residuals"]).max())
        out[f"drift/ep{ep}/weights"] = np.array(np.abs(w64.detach().numpy() - out[f"ep{ep}/weights"]).max())
        out[f"drift/ep{ep}/threshold"] = np.array(abs(float(thr64) - float(out[f"ep{ep}/threshold"])))
    save("g4_epoch", **out)


def gen_g5(ref):
    sys.path.insert(0, os.path.join(ref, "standard-learning"))
    import rlvi as ref_rlvi
    out = {}
    for n in (40, 1000):
        for kind in ("exp", "bimodal", "heavy"):
            l = synth.residual_vector(kind, n, seed=50 + n).astype(np.float64)
            errs = []
            orig = np.linalg.norm

            def spy(*a, **k):
                v = orig(*a, **k)
                errs.append(float(v))
                return v
            np.linalg.norm = spy
            try:
                w = ref_rlvi.update_weights(l)
            finally:
                np.linalg.norm = orig
            out[f"uw_{kind}_{n}/losses"] = l
            out[f"uw_{kind}_{n}/w"] = w
            out[f"uw_{kind}_{n}/errs"] = np.array(errs)
    for (size, d, seed) in ((40, 10, 1), (1000, 20, 0)):
        X, y = synth.linreg_data(size=size, d=d, eps=0.3, nu=2.5, seed=seed)
        calls = []
        orig_uw = ref_rlvi.update_weights

        def spy_uw(losses, **k):
            w = orig_uw(losses, **k)
            calls.append(w)
            return w
        ref_rlvi.update_weights = spy_uw
        try:
            theta = ref_rlvi.linear_regression(X, y)
        finally:
            ref_rlvi.update_weights = orig_uw
        key = f"linreg_{size}x{d}"
        out[key + "/seed"] = np.array(seed)
        out[key + "/theta"] = theta
        out[key + "/outer"] = np.array(len(calls))
        out[key + "/w_last"] = calls[-1]
    # rank-deficient design (two identical columns, one all-zero column): the reference's lstsq
    # (rlvi.py:71,:80, LAPACK gelsd) returns the MINIMUM-NORM least-squares solution
    X, y = synth.linreg_data(size=200, d=6, eps=0.2, nu=2.5, seed=4)
    X = np.concatenate([X, X[:, 2:3], np.zeros((200, 1))], axis=1)
    theta = ref_rlvi.linear_regression(X, y)
    out["linreg_rankdef/X"] = X
    out["linreg_rankdef/y"] = y
    out["linreg_rankdef/theta"] = theta
    from scipy.linalg import lstsq as _lstsq
    wr = np.random.default_rng(5).random(200)
    out["linreg_rankdef/w"] = wr
    out["linreg_rankdef/theta_wls"] = _lstsq(np.sqrt(wr)[:, None] * X, np.sqrt(wr) * y)[0]
    # logistic regression: generate_data_logistic_regression-style 2-D data
    rng = np.random.default_rng(3)
    n = 200
    X = rng.standard_normal((n, 2))
    p = 1 / (1 + np.exp(-(0.5 + 2 * X[:, 0] - X[:, 1])))
    yb = (rng.random(n) < p).astype(np.float64)
    yb[:10] = 1 - yb[:10]
    theta = ref_rlvi.logistic_regression(X.copy(), yb.copy())
    out["logreg/X"] = X
    out["logreg/y"] = yb
    out["logreg/theta"] = theta
    save("g5_standard", **out)


def gen_g6(ref):
    import scipy.io
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    scipy.io.savemat(os.path.join(tmp, "humanactivity.mat"),
                     {"feat": rng.standard_normal((400, 60)),
                      "actid": rng.integers(1, 6, (400, 1))})
    os.chdir(tmp)
    os.environ["MPLBACKEND"] = "Agg"
    sys.path.insert(0, os.path.join(ref, "online-learning"))
    import main as online
    out = {}
    for B in (100, 256):
        for kind in ("exp", "bimodal", "heavy"):
            l = synth.residual_vector(kind, B, seed=60 + B).astype(np.float64)
            w = online.update_weights_rlvi(l.copy())
            out[f"uw_{kind}_{B}/losses"] = l
            out[f"uw_{kind}_{B}/w"] = w
    # first-batch residual: log(0.5) everywhere (main.py:293)
    l0 = -np.log(0.5 * np.ones(100))
    out["uw_first/losses"] = l0
    out["uw_first/w"] = online.update_weights_rlvi(l0.copy())
    lp = np.log(rng.random(50))
    t = rng.integers(0, 2, 50).astype(np.float64)
    out["ce/log_proba"] = lp
    out["ce/targets"] = t
    out["ce/out"] = online.cross_entropy(lp, t)
    os.chdir(ROOT)
    save("g6_online", **out)


def gen_g7(ref):
    """standard-learning estimators on top of the same E-step: mean, pca, covariance
    (rlvi.py:46-65, :111-144) and update_weights_constrained (:23-43)."""
    sys.path.insert(0, os.path.join(ref, "standard-learning"))
    import rlvi as ref_rlvi
    out = {}
    for size, seed in ((60, 2), (200, 1)):
        x = synth.heavy_tail_cloud(size=size, eps=0.2, seed=seed)
        out[f"mean_{size}/theta"] = ref_rlvi.mean(x.copy())
        out[f"pca_{size}/theta"] = ref_rlvi.pca(x.copy())
        out[f"cov_{size}/theta"] = ref_rlvi.covariance(x.copy(), eps=0.4)
        out[f"seed_{size}"] = np.array(seed)
    l = synth.residual_vector("heavy", 100, seed=5).astype(np.float64)
    out["uwc/losses"] = l
    out["uwc/n_eff"] = np.array(60.0)
    out["uwc/w"] = ref_rlvi.update_weights_constrained(l.copy(), 60.0)
    save("g7_estimators", **out)


def gen_g8(ref):
    """Small-loss selection baselines: loss values and logits gradients."""
    import torch
    ref_deep(ref)
    usdnl = sys.modules["methods.train_usdnl"]
    cot = sys.modules["methods.train_coteaching"]
    out = {}
    keys = []
    for (B, C, fr) in ((64, 10, 0.2), (200, 100, 0.45), (1000, 14, 0.0), (37, 10, 0.9)):
        seed = 800 + B + C
        d1 = synth.mstep_inputs(B, C, N=B, seed=seed, zero_frac=0.0)
        d2 = synth.mstep_inputs(B, C, N=B, seed=seed + 1, zero_frac=0.0)
        labels = torch.from_numpy(d1["labels"])
        key = f"B{B}_C{C}"
        keys.append(key)
        out[key + "/B"], out[key + "/C"] = np.array(B), np.array(C)
        out[key + "/seed"], out[key + "/forget_rate"] = np.array(seed), np.array(fr)
        z = torch.from_numpy(d1["logits"].copy()).requires_grad_(True)
        loss = usdnl.loss_fn(z, labels, fr)
        loss.backward()
        out[key + "/usdnl_loss"] = np.array(np.float32(loss.item()))
        out[key + "/usdnl_grad"] = z.grad.numpy()
        z1 = torch.from_numpy(d1["logits"].copy()).requires_grad_(True)
        z2 = torch.from_numpy(d2["logits"].copy()).requires_grad_(True)
        l1, l2 = cot.loss_coteaching(z1, z2, labels, fr, None)
        (l1 + l2).backward()
        out[key + "/cot_loss1"] = np.array(np.float32(l1.item()))
        out[key + "/cot_loss2"] = np.array(np.float32(l2.item()))
        out[key + "/cot_grad1"] = z1.grad.numpy()
        out[key + "/cot_grad2"] = z2.grad.numpy()
    out["cases"] = np.array(keys)
    save("g8_small_loss", **out)


def gen_g9(ref):
    """Top-1 on rows whose maximum is attained more than once: the reference's evaluate()
    (deep-learning/utils.py:48-62: softmax -> torch.max -> eq) on logits handed through an identity
    model.  Values are small integers / halves so that the bf16 path sees the same rows."""
    import torch
    sys.path.insert(0, os.path.join(ref, "deep-learning"))
    import utils as ref_utils
    rng = np.random.default_rng(9)
    out, keys = {}, []
    for C in (10, 100, 101):
        B = 96
        z = rng.integers(-6, 3, (B, C)).astype(np.float32)       # many exact ties below the maximum
        labels = rng.integers(0, C, B).astype(np.int64)
        for i in range(B):
            kind = i % 6
            cols = rng.choice(C, size=1 + (i % 4), replace=False)
            z[i, cols] = 4.0 + 0.5 * (i % 3)                      # the tied maxima
            if kind == 0: labels[i] = cols.min()                  # label = first maximum: hit
            elif kind == 1: labels[i] = cols.max()                # label = last maximum: hit only if single
            elif kind == 2: z[i, :] = 1.5; labels[i] = 0          # constant row, label first
            elif kind == 3: z[i, :] = -2.0; labels[i] = C - 1     # constant row, label last
            elif kind == 4: labels[i] = cols[0]                   # some maximum
            # kind 5: random label
        zt, yt = torch.from_numpy(z), torch.from_numpy(labels)
        loader = [(zt[s:s + 32], yt[s:s + 32], None) for s in range(0, B, 32)]
        acc = ref_utils.evaluate(loader, torch.nn.Identity())
        _, pred = torch.max(torch.nn.functional.softmax(zt, dim=1).data, 1)
        key = f"C{C}"
        keys.append(key)
        out[key + "/logits"] = z
        out[key + "/labels"] = labels
        out[key + "/acc"] = np.array(acc)
        out[key + "/hit"] = (pred == yt).numpy()
    out["cases"] = np.array(keys)
    save("g9_top1_ties", **out)


def gen_g10(ref):
    """precision@k: the reference's accuracy(logit, target, topk=(1, 5)) (deep-learning/utils.py:65-79: softmax ->
    topk -> eq) on seeded logits WITHOUT ties (continuous values; the order topk gives equal values is an
    implementation detail), shapes incl. C = 5 (k = C) and a ragged batch; per-row rank of the label beside it."""
    import torch
    sys.path.insert(0, os.path.join(ref, "deep-learning"))
    import utils as ref_utils
    rng = np.random.default_rng(10)
    out, keys = {}, []
    for B, C in ((32, 10), (128, 100), (777, 101), (64, 5), (40, 1000)):
        z = (3.0 * rng.standard_normal((B, C))).astype(np.float32)
        labels = rng.integers(0, C, B).astype(np.int64)
        clean = rng.random(B) < 0.4
        z[np.nonzero(clean)[0], labels[clean]] += np.float32(4.0)
        zt, yt = torch.from_numpy(z), torch.from_numpy(labels)
        p1, p5 = ref_utils.accuracy(zt, yt, topk=(1, 5))
        p3, = ref_utils.accuracy(zt, yt, topk=(3,))
        key = f"B{B}_C{C}"
        keys.append(key)
        out[key + "/logits"] = z
        out[key + "/labels"] = labels
        out[key + "/prec"] = np.array([float(p1), float(p3), float(p5)], np.float64)
        out[key + "/shape_of_result"] = np.array(p1.shape)
    out["cases"] = np.array(keys)
    save("g10_topk", **out)


GROUPS = {"g10": gen_g10, "g9": gen_g9, "g7": gen_g7, "g8": gen_g8, "g12": gen_g1_g2, "g3": gen_g3, "g4": gen_g4, "g5": gen_g5, "g6": gen_g6}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--group", default=None)
    a = ap.parse_args()
    if a.group:
        GROUPS[a.group](a.ref)
    else:
        # one subprocess per group: deep-learning/ and standard-learning/ both
        # have a top-level `utils` module
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for g in GROUPS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--ref", a.ref,
                                   "--group", g], env=env)
