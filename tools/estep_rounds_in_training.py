"""How many trajectory rounds the per-epoch E-step takes in a real training run (warm start =
last epoch's trajectory, which the moving losses make a poor guess).  Needs RLVI_TJ_DEBUG=1."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rlvi_amd import driver, ops  # noqa: E402
from rlvi_amd.methods import train_rlvi as _fn  # noqa: E402,F401
import rlvi_amd.methods.train_rlvi  # noqa: E402,F401

SCRATCH_OFF = ops.debug_scratch_offset()
mod = sys.modules["rlvi_amd.methods.train_rlvi"]
orig = ops.epoch_end
dev = torch.device("cuda:0")


def spy(residuals, weights, *a, **k):
    ws = k.get("ws") or ops.workspace(weights.device, weights.shape[0], 0)
    ws.buf[SCRATCH_OFF + 64 * 8:SCRATCH_OFF + 88 * 8].zero_()
    it = torch.zeros(1, dtype=torch.int32, device=weights.device)
    k["iters"] = it
    out = orig(residuals, weights, *a, **k)
    torch.cuda.synchronize()
    raw = ws.buf[SCRATCH_OFF + 64 * 8:SCRATCH_OFF + 88 * 8].cpu().numpy().view(np.uint64)
    deltas = [round(float(np.array([x & 0xFFFFFFFF], np.uint32).view(np.float32)[0]), 5) for x in raw if x]
    print(f"  E-step N={weights.shape[0]}: iterations {int(it)}, rounds {len(deltas)}, deltas {deltas}", flush=True)
    return out


ops.epoch_end = spy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for r in driver.run(n_train=n, n_val=2048, n_test=2048, batch_size=1024, n_epoch=9, lr=0.1):
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
