"""Lab: random launches of the M-step (rows 1 ... 70 000 log-uniform, classes 1 ... 300 and a few up to 5000, fp32 /
bf16, dense or padded pitch for logits and gradient, with / without gradient, `out` or accumulate + reduce) against the
oracle.  The launcher has a dozen forms chosen by shape and launch size; this looks for holes between them."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from oracle import rlvi_oracle as oracle  # noqa: E402  (lab: the checker)
from rlvi_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dense_mode = len(sys.argv) > 3 and sys.argv[3] == "dense"
REL = 1e-5
bad = 0
forms = {}
ws = ops.Workspace(dev, 80000, 80000)
for c in range(cases):
    if dense_mode:                                            # (argv[3] = dense) dense launches of 8192 ... 70 000 rows: the tile forms
        B = int(rng.integers(8192, 70001))
        C = int(rng.integers(1, 521))
        if B * C > 12_000_000:
            B = max(8192, 12_000_000 // C)
    elif rng.random() < 0.35:                                 # chip-filling launches: the wave-tile forms
        B = int(rng.integers(20000, 70001))
        C = int(rng.integers(1, 161))
    else:
        B = int(np.exp(rng.uniform(0, np.log(70000))))
        C = int(rng.integers(1, 301)) if rng.random() < 0.9 else int(rng.integers(301, 5001))
    if B * C > 12_000_000:
        B = max(1, 12_000_000 // C)
    dtype = "bf16" if rng.random() < 0.4 else "f32"
    pad_in = 0 if dense_mode else int(rng.choice([0, 0, 1, 3, 4, 8]))
    pad_g = 0 if dense_mode else int(rng.choice([0, 0, 2, 4, 8]))
    want_grad = rng.random() < 0.85
    accumulate = rng.random() < 0.4
    N = B + int(rng.integers(0, 50))
    d = synth.mstep_inputs(B, C, N=N, seed=5000 + c, zero_frac=0.1)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    z = torch.from_numpy(d["logits"]).to(dev).to(tdt)
    if dtype == "bf16":
        d["logits"] = z.float().cpu().numpy()
    if pad_in:
        zp = torch.zeros((B, C + pad_in), device=dev, dtype=tdt)
        zp[:, :C] = z
        z = zp[:, :C]
    grad = None
    if want_grad:
        grad = torch.full((B, C + pad_g), 7.0, device=dev, dtype=tdt)[:, :C]
    res = torch.from_numpy(d["residuals"].copy()).to(dev)
    lab, idx, w = (torch.from_numpy(d[k]).to(dev) for k in ("labels", "idx", "weights"))
    out, g = ops.mstep_fwd_bwd(z, lab, idx, w, res, want_grad=want_grad, grad=grad, ws=ws, accumulate=accumulate)
    if accumulate:
        out = ops.mstep_reduce(ws=ws)
    torch.cuda.synchronize()
    form = _lib.load().rlvi_workspace_last_mstep_form(ws.ptr)
    forms[form] = forms.get(form, 0) + 1
    o = out.cpu().numpy()
    r0 = d["residuals"].copy()
    ref = oracle.mstep(d["logits"], d["labels"], d["idx"], d["weights"], r0, want_grad=want_grad)
    why = []
    if not np.allclose(res.cpu().numpy(), r0, rtol=REL, atol=1e-6):
        why.append("residuals")
    # (+ 2e-7: a batch of one or two clean rows has a loss of 1e-4, i.e. log(1 + 1e-4) in fp32 on both sides)
    if abs(float(o[0]) - float(ref["loss"])) > REL * abs(float(ref["loss"])) + 2e-7:
        why.append(f"loss {o[0]} / {ref['loss']}")
    if float(o[3]) != float(round(float(ref["prec1"]) * B / 100.0)):
        why.append(f"hits {o[3]} / {float(ref['prec1']) * B / 100.0}")
    if want_grad:
        gh = g.float().cpu().numpy()
        if dtype == "bf16":
            rg = torch.from_numpy(ref["grad"].astype(np.float32)).to(torch.bfloat16).float().numpy()
            if not np.allclose(gh, rg, rtol=2 ** -7, atol=1e-7):
                why.append(f"grad (bf16) max abs {np.abs(gh - rg).max():.3e}")
        else:
            diff = gh.astype(np.float64) - ref["grad"]
            # (a batch of one or two clean rows: every entry is p - onehot at p = 0.9999, good to one fp32 ulp of 1 on both sides)
            fro_bad = np.sqrt((diff ** 2).sum()) > REL * max(np.sqrt((ref["grad"].astype(np.float64) ** 2).sum()), 1e-12)
            if (fro_bad and np.abs(diff).max() > 1.2e-7) or np.abs(diff).max() > 1e-6:
                why.append(f"grad max abs {np.abs(diff).max():.3e}")
        if pad_g and not bool((grad.storage_offset() == 0) and (torch.full((B, C + pad_g), 7.0, device=dev, dtype=tdt)[:, C:] == g.as_strided((B, pad_g), (C + pad_g, 1), C)).all()):
            why.append("the gradient's padding was written")
    st = ws.status()
    if st:
        why.append(f"status {st}")
        ws.clear_status()
    if why:
        bad += 1
        print(f"case {c}: B={B} C={C} {dtype} pad_in={pad_in} pad_g={pad_g} grad={want_grad} accumulate={accumulate} form={form}: " + "; ".join(why))
print(f"{cases} cases: {bad} disagreements; forms taken {dict(sorted(forms.items()))}")
