"""Lab: random training set-ups (N, batch size incl. a ragged last batch and batches of one row, classes, features,
optimizer, overfit pattern, a subset of the indices never visited) through the plug-in train_rlvi against the test
suite's checker epoch (stock torch batch loop + the pinned C oracle at the epoch end) from the same initial state:
train_acc, threshold, residuals, weights, parameters after every epoch.  Tolerances are loose (1e-3 relative: two
fp32 runs of a training loop drift apart -- the SGD step is scaled down for batches of a few rows, where 1400 steps of
momentum SGD per epoch amplify one ulp to a per cent within two epochs): this looks for structural slips, not ulps."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from rlvi_amd import ops  # noqa: E402
from rlvi_amd.methods.train_rlvi import train_rlvi  # noqa: E402
from test_gpu_parity import _eager_train_rlvi  # noqa: E402  (lab: the checker)

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for c in range(cases):
    N = int(np.exp(rng.uniform(np.log(8), np.log(6000))))
    B = int(rng.choice([1, 2, 7, 32, 100, 128, 500, 4096]))
    C = int(rng.choice([2, 3, 5, 10, 17, 100, 101]))
    F_ = int(rng.integers(4, 40))
    visit = N if rng.random() < 0.7 else max(1, int(N * 0.9))       # food.py's quirk: slots that are never visited
    X = torch.from_numpy(rng.standard_normal((N, F_)).astype(np.float32))
    Wt = rng.standard_normal((F_, C)).astype(np.float32)
    y_clean = (X.numpy() @ Wt).argmax(1)
    flip = rng.random(N) < 0.3
    y = torch.from_numpy(np.where(flip, rng.integers(0, C, N), y_clean).astype(np.int64))
    torch.manual_seed(100 + c)
    model0 = torch.nn.Linear(F_, C).to(dev)
    opt_kind = "sgd" if rng.random() < 0.6 else "adam"
    pattern = [False, bool(rng.random() < 0.5), True, True]
    if B == 1 and visit > 300:
        visit = 300

    def loaders(seed):
        out = []
        for ep in range(len(pattern)):
            perm = np.random.default_rng(seed + ep).permutation(visit)
            out.append([(X[perm[s:s + B]], y[perm[s:s + B]], torch.from_numpy(perm[s:s + B])) for s in range(0, visit, B)])
        return out

    states = []
    for which in ("product", "checker"):
        model = copy.deepcopy(model0)
        opt = torch.optim.SGD(model.parameters(), lr=0.05 * min(1.0, B / 32.0), momentum=0.9) if opt_kind == "sgd" else torch.optim.Adam(model.parameters(), lr=1e-3)
        residuals = torch.zeros(N, device=dev)
        weights = torch.ones(N, device=dev)
        thr = 0
        hist = []
        for ep, loader in enumerate(loaders(7 * c)):
            fn = train_rlvi if which == "product" else _eager_train_rlvi
            acc, thr = fn(loader, model, opt, residuals, weights, pattern[ep], thr)
            hist.append((float(acc), float(thr), residuals.detach().cpu().numpy().copy(), weights.detach().cpu().numpy().copy(),
                         torch.cat([p.detach().flatten() for p in model.parameters()]).cpu().numpy().copy()))
        states.append(hist)
    why = []
    for ep, (a, b) in enumerate(zip(*states)):
        if abs(a[0] - b[0]) > 100.0 / min(B, visit) * 0.5 + 1e-3:
            why.append(f"epoch {ep}: train_acc {a[0]:.4f} / {b[0]:.4f}")
        if abs(a[1] - b[1]) > 2e-3 * max(abs(b[1]), 1e-3):
            why.append(f"epoch {ep}: threshold {a[1]:.6g} / {b[1]:.6g}")
        for name, i, tol in (("residuals", 2, 2e-3), ("weights", 3, 5e-3), ("parameters", 4, 2e-3)):
            d = float(np.max(np.abs(a[i] - b[i])) / max(float(np.max(np.abs(b[i]))), 1e-6))
            # a truncated weight is 0 on one side and just above the threshold on the other when the two runs' weights
            # straddle it: count such entries instead
            if name == "weights" and d > tol:
                strad = int(((a[i] == 0) != (b[i] == 0)).sum())
                rest = (a[i] == 0) == (b[i] == 0)
                d2 = float(np.max(np.abs(a[i][rest] - b[i][rest]))) if rest.any() else 0.0
                if strad > max(2, N // 200) or d2 > tol:
                    why.append(f"epoch {ep}: weights differ (straddling the threshold: {strad}, others max abs {d2:.2e})")
            elif d > tol:
                why.append(f"epoch {ep}: {name} rel {d:.2e}")
    st = ops.workspace(dev).status()
    if st:
        why.append(f"status {st}")
        ops.workspace(dev).clear_status()
    if why:
        bad += 1
        print(f"case {c}: N={N} visited={visit} B={B} C={C} F={F_} {opt_kind} overfit={pattern}: " + "; ".join(why[:4]))
print(f"{cases} cases: {bad} disagreements")
