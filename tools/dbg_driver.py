import sys; sys.path.insert(0, ".")
import numpy as np, torch
from rlvi_amd import driver, ops
from rlvi_amd.methods import train_rlvi
import rlvi_amd.methods.train_rlvi as T
dev = torch.device("cuda:0")
orig = ops.epoch_end
saved = []
def spy(residuals, weights, **kw):
    r0 = residuals.clone(); w0 = weights.clone()
    out = orig(residuals, weights, **kw)
    torch.cuda.synchronize()
    st = kw["ws"].status()
    saved.append((r0.cpu().numpy(), w0.cpu().numpy(), weights.cpu().numpy().copy(), st))
    print("epoch_end status", st, "res min/max", float(r0.min()), float(r0.max()))
    if st:
        from rlvi_amd import _lib
        _lib.load().rlvi_workspace_init(kw["ws"].ptr, kw["ws"].nbytes, None)
    return out
ops.epoch_end = spy
logs = driver.run(n_train=8192, n_val=1024, n_test=2048, batch_size=1024, n_epoch=9, lr=0.1)
np.savez_compressed("gpurun_out/driver_dbg.npz", **{f"r{i}": s[0] for i, s in enumerate(saved)}, **{f"w{i}": s[1] for i, s in enumerate(saved)}, **{f"o{i}": s[2] for i, s in enumerate(saved)}, st=np.array([s[3] for s in saved]))
