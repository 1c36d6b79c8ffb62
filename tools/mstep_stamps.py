#!/usr/bin/env python3
"""Phase timeline of the wave-tile M-step kernel from a -DRLVI_MSTEP_STAMPS build.
    RLVI_LIB_PATH=rlvi_amd/librlvi_stamps.so python tools/mstep_stamps.py [--tune NAME=V ...]
Stamps per wave (first tile): 0 start, 1 DMA issued, 2 tile landed, 3 row max, 4 row sum,
5 arithmetic done, 6 stores issued, 7 stores retired.  Prints percentiles in us since the first wave's start."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rlvi_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tune", action="append", default=[])
ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--classes", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, C = a.rows, a.classes
d0, labels, idx, logits, grads, weights, residuals = bench.make_inputs(torch, dev, B, C, B, 0)
for kv in a.tune:
    k, v = kv.split("=")
    _lib.check(_lib.load().rlvi_tune_set(k.encode(), int(v)), "tune")
ws = ops.Workspace(dev, B, B)
for i in range(24):
    ops.mstep_fwd_bwd(logits[i % 12], labels, idx, weights, residuals, grad=grads[i % 12], ws=ws, accumulate=True)
torch.cuda.synchronize()
off = ops.debug_scratch_offset()
raw = ws.buf[off:off + 4096 * 128].cpu().numpy().view(np.uint64).reshape(-1, 16).astype(np.int64)
raw = raw[raw[:, 0] > 0]
if len(raw) == 0:
    sys.exit("no stamps in the workspace: this library was not built with -DRLVI_MSTEP_STAMPS\n"
             "  python tools/build_variants.py stamps=-DRLVI_MSTEP_STAMPS\n"
             "  RLVI_LIB_PATH=rlvi_amd/librlvi_stamps.so python tools/mstep_stamps.py")
t0 = raw[:, 0].min()
names = ["start", "dma issued", "tile landed", "row max", "row sum", "arith done", "stores issued", "stores retired"]
print(f"{len(raw)} waves; us since first wave start (min / p10 / median / p90 / max)")
for k, nm in enumerate(names):
    x = (raw[:, k] - t0) / 100.0
    print(f"  {nm:15s} {x.min():6.2f} {np.percentile(x, 10):6.2f} {np.median(x):6.2f} {np.percentile(x, 90):6.2f} {x.max():6.2f}")
d = np.diff(raw[:, :8], axis=1) / 100.0
print("phase durations (median / p90): " + ", ".join(f"{names[k+1]} {np.median(d[:, k]):.2f}/{np.percentile(d[:, k], 90):.2f}" for k in range(7)))

# placement: HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9 layout); XCC_ID bits 3:0
hw, xcc = raw[:, 8], raw[:, 9] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 10 * 2 + cu
import collections
cnt = collections.Counter(key.tolist())
print(f"distinct CUs seen: {len(cnt)}; waves per CU histogram: {sorted(collections.Counter(cnt.values()).items())}")
end = {}
for kk, e in zip(key.tolist(), ((raw[:, 7] - t0) / 100.0).tolist()):
    end[kk] = max(end.get(kk, 0.0), e)
by = collections.defaultdict(list)
for kk, e in end.items():
    by[cnt[kk]].append(e)
for n in sorted(by):
    v = np.array(by[n])
    print(f"  CUs with {n:2d} waves: {len(v):3d}  last store retired median {np.median(v):.2f} max {v.max():.2f}")
print("raw hw_id sample:", [hex(int(x)) for x in hw[:6]], "xcc:", xcc[:16].tolist())
land = (raw[:, 2] - t0) / 100.0
fin = (raw[:, 7] - t0) / 100.0
iss = (raw[:, 1] - t0) / 100.0
print("per XCD: waves, dma-issued median/max, tile-landed median/max, retired median/max")
for x in range(8):
    mk = xcc == x
    print(f"  xcc {x}: {mk.sum():4d}  {np.median(iss[mk]):5.2f}/{iss[mk].max():5.2f}  {np.median(land[mk]):5.2f}/{land[mk].max():5.2f}  {np.median(fin[mk]):5.2f}/{fin[mk].max():5.2f}")
tile = np.arange(len(raw))
print("by tile index (16 bins): landed median / max")
for b in range(16):
    mk = (tile * 16 // len(raw)) == b
    print(f"  bin {b:2d}: {np.median(land[mk]):5.2f} / {land[mk].max():5.2f}   issued {np.median(iss[mk]):5.2f} / {iss[mk].max():5.2f}")
# within a CU: order of issue
simd = (hw >> 4) & 0x3
wslot = hw & 0xF
print("by wave slot id within SIMD (hw wave id): issued median, landed median")
for w in sorted(set(wslot.tolist())):
    mk = wslot == w
    print(f"  wave_id {w:2d}: n={mk.sum():4d} issued {np.median(iss[mk]):5.2f} landed {np.median(land[mk]):5.2f} retired {np.median(fin[mk]):5.2f}")
