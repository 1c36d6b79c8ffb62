"""Lab: random vectors of random length (1 ... 2 000 000, log-uniform) through the type-II threshold, the truncation and
the mask against the oracle -- bit for bit; one workspace per size class (every call's guesses come from an unrelated
vector), alpha and the previous threshold vary; the epoch end (E-step + threshold in one call) on some of them."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from oracle import rlvi_oracle as oracle  # noqa: E402  (lab: the checker)
from rlvi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
wss = {}
for c in range(cases):
    N = int(np.exp(rng.uniform(0, np.log(2_000_000))))
    kind = int(rng.integers(0, 8))
    u = rng.random(N)
    if kind == 0:
        w = u
    elif kind == 1:
        w = 1.0 - u ** float(rng.choice([2, 4, 8]))
    elif kind == 2:
        w = u ** float(rng.choice([2, 4, 8]))
    elif kind == 3:
        q = float(2 ** int(rng.integers(1, 21)))
        w = np.round(u * q) / q
    elif kind == 4:
        w = np.sort(u)[::int(rng.choice([1, -1]))]
    elif kind == 5:
        w = np.where(u < 0.3, 0.0, np.where(u > 0.8, 1.0, rng.random(N)))
    elif kind == 6:
        w = np.full(N, float(rng.choice([0.0, 1.0, 0.37])))
    else:
        w = u * 1.5 - 0.2                                          # outside [0, 1]: the generic fp64 form
    w = np.ascontiguousarray(w, dtype=np.float32)
    alpha = float(rng.choice([0.05, 0.05, 0.01, 0.2, 0.5, 0.0]))
    prev = float(rng.choice([0.0, 0.0, 0.5, 0.99]))
    cls = int(np.log2(N)) // 4
    ws = wss.setdefault(cls, ops.Workspace(dev, 2_000_000, 0))
    thr_ref = oracle.false_negative_criterion(w, alpha=alpha)
    expect = max(np.float32(prev), thr_ref)
    w2 = w.copy()
    m_ref = oracle.truncate(w2, expect)
    wt = torch.from_numpy(w.copy()).to(dev)
    t1 = float(ops.fn_threshold(wt, alpha=alpha, ws=ws))
    thr2, mask, kept = ops.threshold_truncate(wt, prev, alpha=alpha, want_mask=True, ws=ws)
    torch.cuda.synchronize()
    why = []
    if t1 != float(thr_ref):
        why.append(f"criterion {t1!r} / {float(thr_ref)!r}")
    if float(thr2) != float(expect):
        why.append(f"threshold {float(thr2)!r} / {float(expect)!r}")
    if not np.array_equal(wt.cpu().numpy(), w2):
        why.append("truncated vector")
    if not np.array_equal(mask.cpu().numpy(), m_ref) or int(kept) != int(m_ref.sum()):
        why.append("mask / kept")
    st = ws.status()
    if st:
        why.append(f"status {st}")
        ws.clear_status()
    if why:
        bad += 1
        print(f"case {c}: N={N} kind={kind} alpha={alpha} prev={prev}: " + "; ".join(why))
print(f"{cases} cases: {bad} disagreements")
