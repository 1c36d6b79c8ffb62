#!/usr/bin/env python3
"""Times single legs of the hot path (HIP events, hipGraph of K launches, rotating buffers).
Used for kernel tuning sweeps:  RLVI_MSTEP_U=2 python tools/time_parts.py --what mstep"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rlvi_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="mstep", choices=["mstep", "mstep_warm", "mstep_out", "estep", "thr", "thr_fn", "fused", "mstep_fwd", "step"])
ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--classes", type=int, default=100)
ap.add_argument("--n", type=int, default=0, help="E-step / threshold vector length (default rows)")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--tag", default="")
ap.add_argument("--seq-idx", action="store_true", help="sample indices in order instead of a permutation (lab: what the random gather / scatter costs)")
ap.add_argument("--tune", action="append", default=[], help="NAME=VALUE knob (rlvi_tune_set); repeatable")
ap.add_argument("--sweep", default="", help="NAME=v1,v2,...: time the leg once per value")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, C = a.rows, a.classes
N = a.n or B
d0, labels, idx, logits, grads, weights, residuals = bench.make_inputs(torch, dev, B, C, B, 0)
if a.seq_idx:
    idx = torch.arange(B, device=dev, dtype=torch.int64)
if a.dtype == "bf16":
    logits = [z.to(torch.bfloat16) for z in logits]
    grads = [g.to(torch.bfloat16) for g in grads]
if N != B:
    from rlvi_amd import synth
    residuals_n = torch.from_numpy(synth.residual_vector("bimodal", N, 1)).to(dev)
    weights_n = torch.rand(N, device=dev) if a.what in ("thr", "thr_fn") else torch.ones(N, device=dev)
out = torch.empty(4, device=dev)
iters = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ws = ops.Workspace(dev, max(N, B), B)
    if N == B:
        ops.mstep_fwd_bwd(logits[0], labels, idx, weights, residuals, out=out, grad=grads[0], ws=ws)
        res_src = residuals.clone()
    else:
        res_src = residuals_n.clone()
        residuals, weights = residuals_n, weights_n
    pi = torch.ones(B, device=dev)
    rows = torch.empty(B, device=dev)
    thr = torch.zeros(1, device=dev)

    def leg(i):
        r = i % bench.ROTATE
        if a.what == "mstep":
            ops.mstep_fwd_bwd(logits[r], labels, idx, weights, residuals, grad=grads[r], ws=ws, accumulate=True)
        elif a.what == "mstep_warm":       # one buffer pair: the block stays in the Infinity Cache
            ops.mstep_fwd_bwd(logits[0], labels, idx, weights, residuals, grad=grads[0], ws=ws, accumulate=True)
        elif a.what == "mstep_out":
            ops.mstep_fwd_bwd(logits[r], labels, idx, weights, residuals, out=out, grad=grads[r], ws=ws)
        elif a.what == "step":
            ops.mstep_fwd_bwd(logits[r], labels, idx, weights, residuals, grad=grads[r], ws=ws, accumulate=True)
            ops.epoch_end(residuals, weights, batches=1, out=out, iters=iters, ws=ws)
        elif a.what == "mstep_fwd":
            ops.mstep_fwd_bwd(logits[r], labels, idx, weights, residuals, out=out, want_grad=False, ws=ws)
        elif a.what == "estep":
            residuals.copy_(res_src)
            ops.estep_deep(residuals, weights, iters=iters, ws=ws)
        elif a.what == "thr":
            from rlvi_amd import _lib
            L = _lib.load()
            _lib.check(L.rlvi_threshold_truncate_f32(ops._ptr(weights), weights.shape[0], 0.05, ops._ptr(thr),
                                                     None, None, ws.ptr, ops._stream_ptr()), "thr")
        elif a.what == "thr_fn":           # the criterion alone: the vector is left as it is (no zeros from a truncation)
            from rlvi_amd import _lib
            L = _lib.load()
            _lib.check(L.rlvi_fn_threshold_f32(ops._ptr(weights), weights.shape[0], 0.05, ops._ptr(thr),
                                               ws.ptr, ops._stream_ptr()), "thr")
        elif a.what == "fused":
            ops.fused_em(logits[r], labels, pi, ws=ws, out=out, grad=grads[r], rows=rows, iters=iters)
    from rlvi_amd import _lib as _L
    for kv in a.tune:
        k, v = kv.split("=")
        _L.check(_L.load().rlvi_tune_set(k.encode(), int(v)), "rlvi_tune_set")

    def time_leg():
        for i in range(12):
            leg(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(a.steps):
                leg(i)
        torch.cuda.synchronize()
        best = 1e9
        reps = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) / a.steps * 1e3)
            best = min(best, reps[-1])
        print("   reps:", " ".join(f"{r:.2f}" for r in reps), flush=True)
        return best

    def describe(best):
        extra = ""
        if a.what in ("mstep", "mstep_warm", "mstep_fwd", "mstep_out"):
            s = 2 if a.dtype == "bf16" else 4
            byt = B * ((1 if a.what == "mstep_fwd" else 2) * C * s + 24)
            extra = f" {byt / best / 1e3:8.1f} GB/s  frac {byt / best / 1e3 / 8000:.3f}"
        if a.what in ("estep", "fused", "step"):
            extra = f" iters {int(iters)}"
        return extra

    if a.sweep:
        name, vals = a.sweep.split("=")
        for v in vals.split(","):
            _L.check(_L.load().rlvi_tune_set(name.encode(), int(v)), "rlvi_tune_set")
            best = time_leg()
            print(f"{a.tag or a.what:20s} {name}={v:>5s} B={B} C={C} N={N} {a.dtype}: {best:8.2f} us/launch"
                  f"{describe(best)}  status={ws.status()}", flush=True)
        sys.exit(0)
    best = time_leg()
    extra = describe(best)
    if any(t.startswith("RLVI_MSTEP_AUTO=1") for t in a.tune):
        import numpy as np
        hs = ws.buf[768:768 + 64].cpu().numpy().view(np.uint64).reshape(4, 2)
        print("   hold slots (key, ticks of 10 ns):", [(hex(int(k)), int(v) & 0xFFFFFF) for k, v in hs if k])
    if a.what in ("thr", "thr_fn") and any(t.startswith("RLVI_THR_DEBUG") for t in a.tune):
        import numpy as np
        off = ops.debug_scratch_offset() + 256
        raw = ws.buf[off:off + 64 * 8].cpu().numpy().view(np.uint64)
        n = int(raw[63])
        st = raw[:n].astype(np.int64)
        print("thr stamps (us since kernel start):", [round(float(x - st[0]) / 100.0, 2) for x in st])
    if os.environ.get("RLVI_TJ_DEBUG"):
        import numpy as np
        from rlvi_amd import _lib
        off = ops.debug_scratch_offset()
        raw = ws.buf[off:off + 240 * 8].cpu().numpy().view(np.uint64)
        n = int(raw[63])
        st = raw[:n].astype(np.int64)
        print("stamps (us since kernel start):", [round(float(x - st[0]) / 100.0, 2) for x in st][:16])
        if a.what == "fused":
            fs = ws.buf[off + 970 * 8:off + 978 * 8].cpu().numpy().view(np.uint64).astype(np.int64)
            print("fused stamps [start, tiles landed, rows done, barrier, solved, grad issued, -, out] us:",
                  [round(float(x - fs[0]) / 100.0, 2) if x else None for x in fs], " solve starts at", round(float(st[0] - fs[0]) / 100.0, 2))
        print('finite mask %x scale' % int(raw[102]), np.array([raw[103] & 0xFFFFFFFF], np.uint32).view(np.float32), 'totals lanes0-7 (S,P,D):', raw[104:128].view(np.float64).reshape(8,3))
        print('round_ok', int(raw[128]), 'it', int(raw[129] >> 32), 'delta', np.array([raw[129] & 0xFFFFFFFF], np.uint32).view(np.float32))
        nn = raw[130:130+44]
        print('rn  :', np.array(nn >> 32, np.uint32).view(np.float32)[:44])
        print('rnew:', np.array(nn & 0xFFFFFFFF, np.uint32).view(np.float32)[:44])
        for xs in range(3):
            rs = raw[200 + xs * 8: 200 + xs * 8 + 7].astype(np.int64)
            print('  exchange', xs, '[stage1->, publish, polled, barrier, totals, chain, epilogue] us since kernel start:', [round(float(v - st[0]) / 100.0, 2) for v in rs])
        rd = raw[64:88]
        print('min, rfin:', np.array([raw[100] & 0xFFFFFFFF, raw[101] & 0xFFFFFFFF], np.uint32).view(np.float32))
        full = ws.buf[off:off + 1000 * 8].cpu().numpy().view(np.uint64)
        print("  wmin stored (wg0):", np.array([full[898] & 0xFFFFFFFF], np.uint32).view(np.float32), " totals val[4] lanes 0..3:", np.array(full[900:904] & 0xFFFFFFFF, np.uint32).view(np.float32), "nq", int(full[900] >> 32))
        so = full[828:892]
        print("  s_k    :", np.array(so >> 32, np.uint32).view(np.float32)[:24])
        print("  rho_k  :", np.array(so & 0xFFFFFFFF, np.uint32).view(np.float32)[:24])
        acc = full[699]
        print("accept-without-verification:", int(acc >> 32), "it", int(acc & 0xFFFFFFFF))
        eb = full[700:764]
        print("  err_est:", np.array(eb >> 32, np.uint32).view(np.float32)[:24])
        print("  band   :", np.array(eb & 0xFFFFFFFF, np.uint32).view(np.float32)[:24])
        ee = full[764:828]
        print("  E_k    :", np.array(ee >> 32, np.uint32).view(np.float32)[:24])
        print("  slope  :", np.array(ee & 0xFFFFFFFF, np.uint32).view(np.float32)[:24])
        cs = full[990:996].astype(np.int64)
        print("recurrence wave [entry, prepared, chain, trust/tail, acceptance, out] us:", [round(float(v - cs[0]) / 100.0, 2) for v in cs])
        if int(full[980]) > 0:
            print("sums of workgroup 0 / wave 0: arithmetic done at", round(float(int(full[980]) - int(st[0])) / 100.0, 2),
                  "butterflies done at", round(float(int(full[981]) - int(st[0])) / 100.0, 2), "us since kernel start")
        if int(full[984]) > 0 and int(full[980]) > int(full[985]):
            print(f"   shader clock from the staged slice to the end of the sums' arithmetic: "
                  f"{(int(full[984]) - int(full[983])) / (int(full[980]) - int(full[985])) * 0.1:.2f} GHz")
        if int(full[961]) > 0:
            print(f"shader clock in the E-step kernel: {int(full[960]) / int(full[961]) * 0.1:.2f} GHz "
                  f"({int(full[960])} cycles in {int(full[961]) / 100.0:.2f} us)")
        print("rounds (it, delta):", [(int(x >> 32), float(np.array([x & 0xFFFFFFFF], np.uint32).view(np.float32)[0])) for x in rd if x][:12])
    print(f"{a.tag or a.what:28s} B={B} C={C} N={N} {a.dtype}: {best:8.2f} us/launch{extra}  status={ws.status()}")
