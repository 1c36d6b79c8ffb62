import sys, os; sys.path.insert(0, ".")
import numpy as np, torch
from rlvi_amd import ops
d = np.load("tools/data/driver_dbg.npz")
dev = torch.device("cuda:0")
i = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ws = ops.Workspace(dev, 8192, 0)
# reproduce the warm state: run the previous epochs first
for j in range(0, i + 1):
    r = torch.from_numpy(d[f"r{j}"].copy()).to(dev); w = torch.from_numpy(d[f"w{j}"].copy()).to(dev)
    it = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.estep_deep(r, w, iters=it, ws=ws)
    torch.cuda.synchronize()
    off = 1024 + 16384 + 32768 + 512 + 32768
    raw = ws.buf[off:off + 240 * 8].cpu().numpy().view(np.uint64)
    rd = raw[64:88]
    print(j, "iters", int(it), "status", ws.status(), "rounds:", [(int(x >> 32), float(np.array([x & 0xFFFFFFFF], np.uint32).view(np.float32)[0])) for x in rd if x][:24])
    ws.buf[off:off+240*8].zero_()
