#!/usr/bin/env python3
"""Reference point for the streaming M-step: torch's float4 copy of the same 26.2 MB block
(read 26.2 MB + write 26.2 MB), same rotation of 12 buffer pairs, same hipGraph timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
B, C, R, K = 65536, 100, 12, 200
dev = torch.device("cuda:0")
src = [torch.randn(B, C, device=dev) for _ in range(R)]
dst = [torch.empty(B, C, device=dev) for _ in range(R)]
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for i in range(R):
        dst[i].copy_(src[i])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for i in range(K):
            dst[i % R].copy_(src[i % R])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side); g.replay(); e1.record(side); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / K * 1e3)
print(f"torch copy 2x{B*C*4/1e6:.1f} MB: {best:.2f} us/launch  {2*B*C*4/best/1e3:.0f} GB/s")
