#!/usr/bin/env python3
"""bench.py -- RLVI samples/sec (fused E+M step) on synthetic logits 65 536 x 100 per GPU.

    python bench.py --gpus N --steps K --warmup W          (spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
                                                           (is one of torchrun's N ranks)

One "step" = one pass of the hot path over one batch, with BOTH halves inside the timed region:
  M  rlvi_mstep_fwd_bwd_f32   per-sample NLL, top-1, residual scatter, lagged-pi gather,
                              pi-weighted loss and dL/dlogits in one streaming pass
                              (reference train_rlvi.py:85-96 without the model)
  E  rlvi_estep_deep_f32      the fixed point over all N samples (train_rlvi.py:14-38 / :99)
i.e. exactly one `train_rlvi` epoch of a one-batch loader, model excluded.  The next step's M
uses this step's pi (lagged pi, as in the reference).  With N ranks every rank streams its own
65 536 rows (weak scaling), the NLL shards are all-gathered (RCCL) and every rank runs the E-step
on the 65 536*N-sample vector (rlvi_amd/dist.py).

Inputs are resident in HBM before the timed region; 12 (logits, grad) buffer pairs rotate so that
> 256 MiB is touched between two uses of a line (HBM-cold, not Infinity-Cache-warm).  The K steps
are captured into one hipGraph so the measurement is not host-launch-bound; timing is HIP events
on the launch stream, bracketed by barrier + synchronize, MAX over ranks.

Extra objects on the JSON line: `roofline` (the streaming M-step kernel in the form train_rlvi runs -- no
caller hint --, algorithmic bytes 2*C*4+24 per sample against 8 TB/s, next to a flat nontemporal copy of the
same bytes in the same rotation and graph: `copy_same_bytes_us`, `frac_of_copy`), `cpu_baseline` (the CPU
oracle timed on this box's host cores), `parts` (per-kernel times: M-step default / hinted / warm / at 4x the
rows, E-step warm / cold / on drifting data, threshold + truncation, the in-batch E+M; `parts.configs`: what
each of BASELINE.json's five configs calls, at its own shape, with the oracle's CPU time beside it),
`parity` (this run's outputs checked against the oracle: loss, gradient, pi, iteration count, selection mask).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
ROTATE = 12                # (logits, grad) pairs: 12 * 52.4 MB = 629 MB > 256 MiB Infinity Cache


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--rows", type=int, default=65536, help="rows per GPU")
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: every rank uses cuda:0 (with --backend gloo on a 1-GPU box)")
    ap.add_argument("--dist-graph", default="off", choices=["auto", "off"],
                    help="world > 1: 'auto' captures the RCCL all-gather in the hipGraph too (and falls "
                         "back to eager launches on every rank if any rank cannot).  Default off: the "
                         "eager loop costs ~37 us of host time per step, less than the GPU needs at "
                         "2+ ranks, and an eager RCCL call cannot hang a replay")
    ap.add_argument("--estep-dist", default="auto", choices=["auto", "sharded", "replicated"],
                    help="world > 1: 'sharded' = every rank keeps its own samples and the E-step kernel exchanges "
                         "its per-node totals through the peers' inboxes (xGMI P2P, no collective call); "
                         "'replicated' = RCCL all-gather of the residuals + the whole E-step on every rank; "
                         "'auto' = sharded if its start-up self-check against the replicated result passes on "
                         "every rank, else replicated")
    ap.add_argument("--force-collective", action="store_true",
                    help="debug: run the residual exchange (and its process group) at world size 1")
    ap.add_argument("--profile-only", action="store_true",
                    help="warmup + timed steps only (for rocprofv3 runs)")
    ap.add_argument("--no-epoch-legs", action="store_true",
                    help="skip the end-to-end epoch timings (parts.epoch: plug-in vs stock eager torch)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="world > 1: weak = --rows rows on EVERY rank (N = rows x world samples); strong = --rows "
                         "rows in all, rows / world on every rank (N = rows samples)")
    ap.add_argument("--debug-absent-peer", type=int, default=-1,
                    help="test hook: this rank never joins the start-up self-check of the sharded E-step, so "
                         "the others' waits run into their bound and every rank lands on the replicated path")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def plan_ranks(requested, visible, same_device):
    """(ranks to start, `not_measured` note or None) for `--gpus requested` on a box with `visible` devices."""
    if same_device or visible >= requested:
        return requested, None
    return max(visible, 0), f"{requested} requested, {visible} visible"


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: become N ranks.  The parent makes NO GPU call (it only
    counts devices, which does not initialise the runtime) and never replaces itself: it starts N fresh
    children -- one per GPU, rendezvous on 127.0.0.1 --, passes rank 0's stdout (the one JSON line) through,
    waits for all of them and exits non-zero if any child does.  Fewer visible devices than asked for (and
    no --same-device): the run is made on the visible ones and says so in the record
    ("not_measured": "N requested, M visible") instead of dressing an M-GPU number as an N-GPU one."""
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    env = dict(os.environ)
    n, note = plan_ranks(a.gpus, visible, a.same_device)
    if note is not None:
        print(f"# bench: {note}", file=sys.stderr)
        if n < 1:
            print(json.dumps({"metric": "RLVI samples/sec (fused E+M step)", "value": None, "unit": "samples/s",
                              "n_gpus": 0, "not_measured": note}))
            return 3
        env["RLVI_BENCH_REQUESTED_GPUS"] = str(a.gpus)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else sys.stderr))
    # wait for all; a rank that fails takes the others with it after a grace period (they would wait in
    # a collective for ever): the exact children, by pid
    rc, t_fail = 0, None
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is not None:
                live.remove(p)
                if c != 0 and rc == 0:
                    rc, t_fail = c, time.time()
        if t_fail is not None and live and time.time() - t_fail > 30.0:
            for p in live:
                p.terminate()
            time.sleep(5.0)
            for p in live:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    return rc



def make_inputs(torch, dev, B, C, N, rank):
    """Recipe of SURVEY 8(d): set 0 is the seeded numpy recipe (used for the parity gate); the
    other rotating sets are the same recipe drawn on the device."""
    import numpy as np
    from rlvi_amd import synth
    d = synth.mstep_inputs(B, C, N=B, seed=synth.BENCH_SEED + rank)
    labels = torch.from_numpy(d["labels"]).to(dev)
    idx_local = torch.from_numpy(d["idx"]).to(dev) + rank * B       # owner-contiguous slices
    logits = [torch.from_numpy(d["logits"]).to(dev)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.BENCH_SEED + 1000 * rank)
    clean = torch.from_numpy(np.random.default_rng(1 + rank).random(B) < 0.55).to(dev)
    rows = torch.arange(B, device=dev)
    for _ in range(ROTATE - 1):
        z = 3.0 * torch.randn((B, C), generator=gen, device=dev, dtype=torch.float32)
        z[rows[clean], labels[clean]] += 12.0
        logits.append(z)
    grads = [torch.empty((B, C), dtype=torch.float32, device=dev) for _ in range(ROTATE)]
    wrng = np.random.default_rng(synth.BENCH_SEED)           # same on every rank: replicated pi
    weights = torch.from_numpy(wrng.random(N).astype(np.float32)).to(dev)
    residuals = torch.zeros(N, dtype=torch.float32, device=dev)
    return d, labels, idx_local, logits, grads, weights, residuals


def eager_torch_epoch(loader, model, optimizer, residuals, weights, tol=1e-3, maxiter=40):
    """What a user of the reference gets on this GPU today: the statements of one train_rlvi epoch
    (deep-learning/methods/train_rlvi.py:79-99 with utils.accuracy, :14-38) as stock PyTorch-ROCm eager ops
    -- written here, no reference file travels -- to stand next to the plug-in's epoch time."""
    import torch
    import torch.nn.functional as F
    hits_pct, batches = 0, 0
    for images, labels, indexes in loader:
        logits = model(images)
        # top-1 of accuracy(logits, labels, topk=(1, 5)): softmax, top-5, compare (the top-5 figure is dropped)
        top = F.softmax(logits, dim=1).topk(5, 1, True, True)[1].t()
        match = top.eq(labels.view(1, -1).expand_as(top))
        hits_pct = hits_pct + match[:1].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / labels.size(0))
        batches += 1
        nll = F.cross_entropy(logits, labels, reduction='none')
        residuals[indexes] = nll
        weighted = (nll * weights[indexes]).mean()
        optimizer.zero_grad()
        weighted.backward()
        optimizer.step()
    with torch.no_grad():          # the per-epoch E-step, one host sync per iteration (the `if error < tol`)
        residuals.sub_(residuals.min())
        e = torch.exp(-residuals)
        avg = 0.95
        for _ in range(maxiter):
            ratio = avg / (1 - avg)
            new = torch.div(ratio * e, 1 + ratio * e)
            err = torch.norm(new - weights)
            weights[:] = new
            avg = weights.mean()
            if err < tol:
                break
        weights.div_(weights.max())
    return float(hits_pct) / float(batches)


def epoch_legs(torch, dev, a):
    """End-to-end epoch wall time through the plug-in -- the one performance quantity the reference itself
    logs (time_ep, deep-learning/main.py:268,327) -- at BASELINE.json's cfg3 shape (LeNet on MNIST-shaped
    synthetic images, N = 54 000) with batch 4096 and with the reference's own batch of 32, next to the
    same epoch as stock eager PyTorch ops on the same GPU, and the host time of the M-step wrapper alone.
    Data is resident on the device (images, labels, indexes); wall clock around a whole epoch including its
    one host sync; median of 3 epochs after one warm-up epoch."""
    from rlvi_amd import driver, ops
    from rlvi_amd.methods import train_rlvi
    N, C = 54000, 10
    out = {}
    x, y, _, _ = driver.synthetic_images(N, 1, C, 28, 0.5, "symmetric", seed=1)
    x, y = x.to(dev), y.to(dev)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(0)
    perm = torch.randperm(N, generator=gen).to(dev)

    def make_loader(bs):
        return [(x[perm[s:s + bs]], y[perm[s:s + bs]], perm[s:s + bs].contiguous()) for s in range(0, N, bs)]

    def run(kind, loader):
        torch.manual_seed(0)
        model = driver.LeNet(1, C).to(dev)
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-3)
        residuals = torch.zeros(N, device=dev)
        weights = torch.ones(N, device=dev)
        model.train()
        times = []
        for ep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if kind == "plugin":
                train_rlvi(loader, model, opt, residuals, weights, False, 0)
            else:
                eager_torch_epoch(loader, model, opt, residuals, weights)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return sorted(times[1:])[1] * 1e3

    for bs in (4096, 32):
        loader = make_loader(bs)
        out[f"b{bs}"] = {"batches": len(loader),
                         "epoch_ms_plugin": run("plugin", loader),
                         "epoch_ms_eager_torch": run("eager", loader)}
        del loader
    # host time of the M-step call alone at cfg3's 4096 x 10 (the kernel takes ~4 us: back-to-back calls on
    # one stream are host-bound, so wall time / calls is what the host pays per batch)
    z = torch.randn(4096, C, device=dev)
    lab = torch.randint(0, C, (4096,), device=dev)
    idx = perm[:4096].contiguous()
    w, r = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    ws = ops.Workspace(dev, N, 4096)
    loop = ops.MStepLoop(w, r, ws)
    for name, fn in (("host_us_per_batch", lambda: loop(z, lab, idx)),
                     ("host_us_per_batch_generic", lambda: ops.mstep_fwd_bwd(z, lab, idx, w, r, ws=ws,
                                                                               accumulate=True))):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3000):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        out[name] = t_host / 3000 * 1e6
    ops.mstep_reduce(ws=ws)
    out["note"] = ("LeNet, N=54000 MNIST-shaped synthetic images resident on the device; wall clock of one whole "
                   "train_rlvi epoch (median of 3 after a warm-up epoch); eager_torch = the same statements "
                   "as stock PyTorch-ROCm ops on this GPU")
    return out


def config_legs(torch, np, dev, ops, timed_part, cpu_jobs, out4):
    """What each of BASELINE.json's five configs calls on this path, at the config's OWN shape (cfg1 ..
    cfg5), each leg a hipGraph of >= 50 calls: `gpu_us` per call.  `cpu_jobs[(cfg, key)] = callable`: the pinned
    oracle on the same inputs, timed in the cpu_baseline leg (`cpu_us`).  The shapes are launches of a few
    microseconds -- launch- and latency-bound, the model's forward / backward dominates a real step there --, so
    these are times, not roofline fractions."""
    import time as _time
    from rlvi_amd import _lib, online, standard, synth
    L = _lib.load()

    class _Oracle:            # the CPU jobs' checker / baseline, imported when the cpu_baseline leg first calls one
        def __getattr__(self, name):
            from oracle import rlvi_oracle
            return getattr(rlvi_oracle, name)
    O = _Oracle()
    cfgs = {}

    def host_call_us(fn, reps=40):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(reps):
            t0 = _time.perf_counter()
            fn()
            ts.append(_time.perf_counter() - t0)
        return float(np.median(ts)) * 1e6

    # ---- cfg1: standard-learning linear regression, n = 1000, d = 20, 30 % outliers (rlvi.py:68-89), fp64:
    # the whole estimator is one launch; `call_us` = standard.linear_regression(X, y) as a numpy caller sees it
    # (pinned H2D of [X | y], the launch, pinned D2H, one host wait)
    X, y = synth.linreg_data(1000, 20, eps=0.3, nu=2.5, seed=0)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    th = torch.empty(20, dtype=torch.float64, device=dev)
    wv = torch.empty(1000, dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)

    def cfg1_leg(i, ws):
        ops.linear_regression(Xd, yd, theta=th, weights=wv, info=info, ws=ws)
    g1 = timed_part(cfg1_leg) * 1e3
    inf = info.cpu().numpy()
    cfgs["cfg1"] = {"what": "linear_regression(X, y) n=1000 d=20 f64, the whole estimator in one launch "
                            "(rlvi_linear_regression_f64)",
                    "gpu_us": g1, "outer_iterations": int(inf[0]), "inner_iterations": int(inf[2]),
                    "call_us": host_call_us(lambda: standard.linear_regression(X, y)),
                    "call_note": "numpy in / numpy out: pinned H2D of [X|y] (168 KB) + launch + pinned D2H + one host wait"}
    cpu_jobs[("cfg1", "cpu_us")] = lambda: O.linear_regression_c(X, y)

    # ---- cfg2: online-learning mini-batch, 256 x 60 (main.py:291-299): X.w -> -log sigmoid -> update_weights_rlvi
    Xl, wl, b = synth.logistic_data(256, 60)
    Xld, wld = torch.from_numpy(Xl).to(dev), torch.from_numpy(wl).to(dev)
    sw = torch.empty(256, dtype=torch.float64, device=dev)
    it2 = torch.zeros(1, dtype=torch.int32, device=dev)

    def cfg2_leg(i, ws):
        ops.sample_weight_online(Xld, wld, b, out=sw, iters=it2)
    g2 = timed_part(cfg2_leg) * 1e3
    cfgs["cfg2"] = {"what": "sample_weight of one online mini-batch 256x60 f64: X.w (MFMA) -> NLL -> online E-step, "
                            "one launch (rlvi_sample_weight_online_f64)",
                    "gpu_us": g2, "estep_iterations": int(it2.item()),
                    "call_us": host_call_us(lambda: online.rlvi_sample_weight(Xl, wl, b))}
    cpu_jobs[("cfg2", "cpu_us")] = lambda: O.update_weights_rlvi(O.logistic_nll(Xl, wl, b))

    # ---- cfg3 / cfg4 / cfg5: the per-batch M-step and the per-epoch end at the deep-learning configs' shapes
    wsc = ops.Workspace(dev, 75750, 4096)
    it_c = torch.zeros(1, dtype=torch.int32, device=dev)
    thr_c = torch.zeros(1, dtype=torch.float32, device=dev)
    walk = [0, 1, 2, 3, 4, 3, 2, 1]
    for name, (Bc, Cc, Nc, bf16, kind, label) in {
            "cfg3": (4096, 10, 54000, False, "bimodal", "MNIST-shaped: batch 4096, 10 classes, N = 54 000"),
            "cfg4": (4096, 10, 45000, False, "bimodal", "CIFAR-10-shaped: 32 768 / 8 ranks = 4096 rows per GPU, "
                                                         "10 classes, N = 45 000"),
            "cfg5": (1024, 101, 75750, True, "zeros10", "Food-101-shaped: 8192 / 8 ranks = 1024 rows per GPU, "
                                                        "101 classes, bf16 logits, N = 75 750 slots (10 % never "
                                                        "visited: residual 0)")}.items():
        d = synth.mstep_inputs(Bc, Cc, N=Nc, seed=3)
        z = torch.from_numpy(d["logits"]).to(dev)
        z_cpu = d["logits"]
        if bf16:
            z = z.to(torch.bfloat16)
            z_cpu = z.float().cpu().numpy()                # the reference fed the bf16-rounded logits as fp32
        lab, idx = torch.from_numpy(d["labels"]).to(dev), torch.from_numpy(d["idx"]).to(dev)
        wts = torch.from_numpy(d["weights"]).to(dev)
        res = torch.zeros(Nc, dtype=torch.float32, device=dev)
        grad = torch.empty_like(z)

        def mstep_leg(i, _ws):
            ops.mstep_fwd_bwd(z, lab, idx, wts, res, grad=grad, ws=wsc, accumulate=True)
        entry = {"what": label, "mstep_rows": Bc, "classes": Cc, "n_samples": Nc,
                 "dtype": "bf16" if bf16 else "f32", "mstep_gpu_us": timed_part(mstep_leg) * 1e3}
        ops.mstep_reduce(ws=wsc)
        w_host = d["weights"]

        def cpu_mstep(z_cpu=z_cpu, d=d, w_host=w_host, Nc=Nc):
            O.mstep(z_cpu, d["labels"], d["idx"], w_host, np.zeros(Nc, np.float32))
        cpu_jobs[(name, "mstep_cpu_us")] = cpu_mstep
        # the epoch end on residuals that move from call to call (a neighbour's trajectory as the guess), without
        # and with the truncation of train_rlvi.py:100-103
        base = synth.residual_vector(kind, Nc, seed=5)
        rng = np.random.default_rng(13)
        dr_np = [(base * np.float32(1.02 ** k) + np.float32(0.01) * rng.random(Nc).astype(np.float32) * (base > 0))
                 .astype(np.float32) for k in range(5)]
        dr = [torch.from_numpy(v).to(dev) for v in dr_np]
        w_e = torch.ones(Nc, dtype=torch.float32, device=dev)

        def end_leg(i, _ws, overfit=0):
            _lib.check(L.rlvi_epoch_end_f32(ops._ptr(dr[walk[i % 8]]), ops._ptr(w_e), Nc, 1e-3, 40, overfit, 0.05,
                                            ops._ptr(thr_c), 0, None, ops._ptr(it_c), wsc.ptr, ops._stream_ptr()),
                       "rlvi_epoch_end_f32")

        def end_trunc_leg(i, _ws):
            end_leg(i, _ws, 1)
        wsc.reset_warm()
        entry["epoch_end_gpu_us"] = timed_part(end_leg) * 1e3
        entry["estep_iterations"] = int(it_c.item())
        thr_c.zero_()
        entry["epoch_end_truncating_gpu_us"] = timed_part(end_trunc_leg) * 1e3

        def cpu_end(v=dr_np[2], Nc=Nc):
            O.update_sample_weights(v.copy(), np.ones(Nc, np.float32))

        def cpu_end_trunc(v=dr_np[2], Nc=Nc):
            w_t = np.ones(Nc, np.float32)
            O.update_sample_weights(v.copy(), w_t)
            O.truncate(w_t, O.false_negative_criterion(w_t))
        cpu_jobs[(name, "epoch_end_cpu_us")] = cpu_end
        cpu_jobs[(name, "epoch_end_truncating_cpu_us")] = cpu_end_trunc
        if name == "cfg3":
            # the in-batch (online-order) E+M at this shape: one launch (a row per thread) / the three-launch composition
            pi_s = torch.ones(Bc, dtype=torch.float32, device=dev)
            rows_s = torch.empty(Bc, dtype=torch.float32, device=dev)
            g_s = torch.empty_like(z)

            def fused_small(i, _ws):
                ops.fused_em(z, lab, pi_s, ws=wsc, out=out4, grad=g_s, rows=rows_s, iters=it_c)
            entry["in_batch_em_gpu_us"] = timed_part(fused_small) * 1e3
            _lib.check(L.rlvi_tune_set(b"RLVI_FUSED_EM", 0), "tune")
            entry["in_batch_em_3launch_gpu_us"] = timed_part(fused_small) * 1e3
            L.rlvi_tune_unset(b"RLVI_FUSED_EM")

            def cpu_v2(z_cpu=z_cpu, d=d, Bc=Bc):
                l_b, _ = O.nll_rows(z_cpu, d["labels"])
                pi_b = np.ones(Bc, np.float32)
                O.update_sample_weights(l_b, pi_b)
                O.mstep(z_cpu, d["labels"], np.arange(Bc), pi_b, np.zeros(Bc, np.float32))
            cpu_jobs[(name, "in_batch_em_cpu_us")] = cpu_v2
        cfgs[name] = entry
    st = wsc.status()
    if st:
        cfgs["device_status"] = st
        wsc.clear_status()
    return cfgs


def main():
    a = parse()
    # IPC handles of device memory (the peers' inboxes of the sharded E-step, RCCL's own buffers) need the
    # dmabuf IPC mode on this driver stack; it has to be in the environment before the runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    import numpy as np
    import torch
    import torch.distributed as dist
    from rlvi_amd import dist as rdist
    from rlvi_amd import ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.same_device:
        local = 0
    not_measured = None
    if "RLVI_BENCH_REQUESTED_GPUS" in os.environ:
        not_measured = f"{os.environ['RLVI_BENCH_REQUESTED_GPUS']} requested, {world} visible"
    elif not a.same_device and torch.cuda.device_count() < world:
        # a launcher made more ranks than there are devices: nothing honest can be measured
        if rank == 0:
            print(json.dumps({"metric": "RLVI samples/sec (fused E+M step)", "value": None, "unit": "samples/s",
                              "n_gpus": torch.cuda.device_count(),
                              "not_measured": f"{world} ranks launched, {torch.cuda.device_count()} devices visible"}))
        sys.exit(3)
    use_dist = world > 1 or a.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        # RCCL prints a version banner on stdout when the communicator is created; stdout is
        # reserved for the one JSON line, so fd 1 points at stderr until the first collective is done
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                dist.barrier(device_ids=[local])
                torch.cuda.synchronize()
            else:
                dist.init_process_group(a.backend)
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        # ranks that drive the same GPU (--same-device) split its co-residency between them
        rdist.declare_device_sharing()
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    if a.gpus != world and rank == 0 and not_measured is None:
        print(f"# note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    C = a.classes
    if a.scaling == "strong" and world > 1:
        if a.rows % (16 * world):
            raise SystemExit(f"--scaling strong: --rows {a.rows} must be a multiple of 16 x {world} ranks")
        B = a.rows // world                  # the same global batch, split over the ranks
    else:
        B = a.rows
    N = B * world
    inv_scale = 1.0 / N                      # global 1/B: per-rank losses / grads SUM to 1 device
    d0, labels, idx_local, logits, grads, weights, residuals = make_inputs(torch, dev, B, C, N, rank)
    out = torch.empty(4, dtype=torch.float32, device=dev)
    iters = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)

    def step(i, ws):
        # what train_rlvi does for a one-batch epoch, model excluded: the fused M-step launch
        # (scalars accumulate on the device) and the epoch end (E-step + scalar reduction)
        r = i % ROTATE
        ops.mstep_fwd_bwd(logits[r], labels, idx_local, weights, residuals, inv_scale=inv_scale,
                          grad=grads[r], ws=ws, accumulate=True)
        epoch_end_any(ws)

    def epoch_end_any(ws):
        if estep_mode[0] == "sharded":
            # every rank keeps its own samples; the kernel's reducers exchange the per-node totals
            ops.estep_sharded(residuals[lo_own:hi_own], weights[lo_own:hi_own], N, batches=1, out=out,
                              iters=iters, ws=ws)
            return
        if use_dist:
            rdist.exchange_residuals_owned(residuals, rank * B, (rank + 1) * B, force=a.force_collective)
        ops.epoch_end(residuals, weights, batches=1, out=out, iters=iters, ws=ws)

    def mstep_only(i, ws):
        r = i % ROTATE
        ops.mstep_fwd_bwd(logits[r], labels, idx_local, weights, residuals, inv_scale=inv_scale,
                          grad=grads[r], ws=ws, accumulate=True)

    def estep_only(i, ws):
        if estep_mode[0] == "sharded":
            epoch_end_any(ws)
        else:
            ops.epoch_end(residuals, weights, batches=1, out=out, iters=iters, ws=ws)

    # The headline step runs the M-step in the form train_rlvi runs it: NO caller hint (round 3 timed the headline
    # under ops.hint_logits_from_hbm, a mode the plug-in never uses).  The hinted form -- every block of the rotation
    # does stream from HBM -- is timed beside it on a workspace of its own (parts.mstep_hinted_us).
    estep_mode = ["replicated" if use_dist else "single"]
    estep_note = ""
    status_log = []        # every non-zero device status seen after a leg: it stays in the JSON line
    last_status = [0]
    lo_own, hi_own = rank * B, (rank + 1) * B
    peers_keep = []

    def agree(flag, note=""):
        """True only if `flag` holds on every rank (any backend)."""
        if not (use_dist and dist.is_initialized()):
            return bool(flag), note
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, (bool(flag), note))
        bad = [f"rank {r}: {n}" for r, (f, n) in enumerate(got) if not f]
        return len(bad) == 0, "; ".join(bad)

    def choose_estep_mode(ws):
        """Set up the peers' inboxes and check one sharded E-step against the replicated one on the same
        vector (status clean, same iteration count, this rank's slice of pi to 1e-5): only then is the
        sharded path used, on every rank or on none."""
        from rlvi_amd import synth
        note, ok = "", True
        try:
            peers_keep.append(rdist.setup_peers(ws))
        except Exception as e:                               # noqa: BLE001 (Peers agreed on it: all ranks)
            return "replicated", f"sharded unavailable ({e})"
        # every rank asks the library whether its sharded launch would be admitted (shape, co-residency
        # this process is entitled to) BEFORE anybody launches: a rank that cannot would leave the others
        # waiting for its records
        from rlvi_amd import _lib as _lc
        can, why = agree(_lc.load().rlvi_estep_sharded_check(B, N, 40, 1) == 0,
                         f"rlvi_estep_sharded_check({B}, {N}) refused")
        if not can:
            return "replicated", f"sharded not launchable ({why})"
        try:
            full = synth.residual_vector("bimodal", N, seed=5)
            r_s = torch.from_numpy(full[lo_own:hi_own].copy()).to(dev)
            w_s = torch.ones(B, dtype=torch.float32, device=dev)
            it_s = torch.zeros(1, dtype=torch.int32, device=dev)
            if rank != a.debug_absent_peer:
                ops.estep_sharded(r_s, w_s, N, iters=it_s, ws=ws)
            ws2 = ops.Workspace(dev, N, 0)
            r_f = torch.from_numpy(full).to(dev)
            w_f = torch.ones(N, dtype=torch.float32, device=dev)
            it_f = torch.zeros(1, dtype=torch.int32, device=dev)
            ops.estep_deep(r_f, w_f, iters=it_f, ws=ws2)
            torch.cuda.synchronize()
            st = ws.status()
            ok = (st == 0 and int(it_s) == int(it_f) and torch.equal(r_s, r_f[lo_own:hi_own]) and
                  torch.allclose(w_s, w_f[lo_own:hi_own], rtol=1e-5, atol=1e-30))
            if not ok:
                note = f"self-check failed (status {st}, iters {int(it_s)} vs {int(it_f)})"
        except Exception as e:                               # noqa: BLE001
            ok, note = False, f"self-check raised {type(e).__name__}: {e}"
        ok, why = agree(ok, note)
        if ok:
            return "sharded", ""
        if ws.status():
            status_log.append({"leg": "sharded self-check", "status": ws.status()})
        ws.clear_status()
        if a.estep_dist == "sharded" and rank == 0:
            print(f"# --estep-dist sharded refused: {why}", file=sys.stderr)
        return "replicated", why

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- parity gate (rank 0, set 0)
    parity = {"checked": False}
    w_before = weights.clone()
    with torch.cuda.stream(side):
        ws = ops.Workspace(dev, N, B)
        torch.cuda.synchronize()
        setup_s = None
        if use_dist and a.estep_dist != "replicated":
            t_setup = time.perf_counter()
            estep_mode[0], estep_note = choose_estep_mode(ws)
            setup_s = time.perf_counter() - t_setup     # peers' set-up + self-check (bounded: see DESIGN 6)
        step(0, ws)
        torch.cuda.synchronize()
        it_gpu = int(iters.item())
    if rank == 0 and world == 1 and not a.profile_only:
        from oracle import rlvi_oracle as O        # the checker, never the thing measured
        res_o = np.zeros(N, np.float32)
        w_o = w_before.cpu().numpy()
        ref = O.mstep(d0["logits"], d0["labels"], d0["idx"], w_o, res_o, scale_div=N)
        it_o = O.update_sample_weights(res_o, w_o)
        gd = grads[0].cpu().numpy().astype(np.float64) - ref["grad"]
        wg = weights.cpu().numpy()
        big = w_o >= 1e-6 * w_o.max()
        parity = {
            "checked": True,
            "loss_rel": abs(float(out[0]) - float(ref["loss"])) / abs(float(ref["loss"])),
            "grad_rel_fro": float(np.sqrt((gd ** 2).sum()) /
                                  np.sqrt((ref["grad"].astype(np.float64) ** 2).sum())),
            "pi_rel_max": float(np.max(np.abs(wg[big] - w_o[big]) / w_o[big])),
            "estep_iters": [it_gpu, int(it_o)],
        }
        # selection mask (train_rlvi.py:41-49,:102-103, main.py:343): threshold + truncation on the
        # GPU's own pi against the oracle on the same vector -- packed bits must be equal
        w_thr = weights.clone()
        thr_g, mask_g, kept_g = ops.threshold_truncate(w_thr, 0.0, want_mask=True, ws=ws)
        torch.cuda.synchronize()
        thr_o = O.false_negative_criterion(wg)
        w_t = wg.copy()
        mask_o = O.truncate(w_t, thr_o)
        parity["mask_bits_equal"] = bool(np.array_equal(np.packbits(mask_g.cpu().numpy()), np.packbits(mask_o)))
        parity["threshold"] = [float(thr_g), float(thr_o)]
        parity["kept"] = [int(kept_g), int(mask_o.sum())]
        parity["ok"] = bool(parity["loss_rel"] <= 1e-5 and parity["grad_rel_fro"] <= 1e-5 and
                            parity["pi_rel_max"] <= 1e-5 and it_gpu == it_o and
                            parity["mask_bits_equal"] and float(thr_g) == float(thr_o) and
                            np.array_equal(w_thr.cpu().numpy(), w_t))

    # ---------------------------------------------------------------- timed regions
    def timed(fn, K, W, use_graph):
        """W untimed + exactly K timed invocations on the side stream; returns ms (max over ranks)."""
        with torch.cuda.stream(side):
            for i in range(W):
                fn(i, ws)
            torch.cuda.synchronize()
            graph = None
            if use_graph:
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        for i in range(K):
                            fn(W + i, ws)
                except Exception as e:              # only the collective can refuse a capture
                    if not use_dist:
                        raise
                    print(f"# rank {rank}: graph capture refused ({type(e).__name__}: {e}); eager",
                          file=sys.stderr)
                    graph = None
                torch.cuda.synchronize()
                if use_dist:                        # every rank takes the same road
                    okf = torch.tensor([1 if graph is not None else 0], dtype=torch.int32,
                                       device=dev if a.backend == "nccl" else "cpu")
                    dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                    if int(okf.item()) == 0:
                        graph = None
                launch_mode[fn.__name__] = "hipGraph" if graph is not None else "eager"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync_all()
            e0.record(side)
            if graph is not None:
                graph.replay()
            else:
                for i in range(K):
                    fn(W + i, ws)
            e1.record(side)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
        sync_all()
        if use_dist:
            t = torch.tensor([ms], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        # a device-side condition raised inside this leg (a wait that hit its bound, a row out of range) is
        # attributed to it, kept for the JSON line and cleared so that the next leg starts clean
        last_status[0] = ws.status()
        if last_status[0]:
            status_log.append({"leg": fn.__name__, "rank": rank, "status": last_status[0],
                               "estep_dist": estep_mode[0]})
            ws.clear_status()
        return ms

    # world > 1: the RCCL all-gather is captured too when RCCL allows it (--dist-graph auto)
    # (the sharded step has no collective call in it: kernels only, captured like the one-GPU step)
    use_graph = (not a.no_graph) and (not use_dist or a.dist_graph == "auto" or estep_mode[0] == "sharded")
    launch_mode = {}
    K, W = a.steps, a.warmup
    ms_total = timed(step, K, W, use_graph)
    if estep_mode[0] == "sharded":
        # a timed region that left a device status behind on any rank (a wait on a peer gave up) is not a
        # measurement: every rank goes back to the replicated E-step and the region is timed again
        clean, why = agree(last_status[0] == 0, f"device status {last_status[0]}")
        if not clean:
            if not status_log or status_log[-1].get("leg") != "step":
                status_log.append({"leg": "step", "rank": rank, "status": 0, "estep_dist": "sharded",
                                   "other_ranks": why})
            estep_mode[0] = "replicated"
            estep_note = f"sharded step left a device status behind ({why}); re-timed replicated"
            use_graph = (not a.no_graph) and a.dist_graph == "auto"
            ms_total = timed(step, K, W, use_graph)
    ms_step = ms_total / K
    value = (B * world) / (ms_step * 1e-3)
    # which physical GPUs ran: the PCI bus id of every rank's device (the same id in every process whatever the
    # visible-device masks made of the ordinals).  Ranks that share a device are a protocol check, never an
    # N-GPU measurement: n_gpus is the number of DISTINCT devices and the line says not_measured.
    import ctypes as _ct
    from rlvi_amd import _lib as _lb
    _buf = _ct.create_string_buffer(64)
    my_bus = _buf.value.decode() if _lb.load().rlvi_device_pci_bus_id(_buf, 64) == 0 else f"unknown-{rank}"
    devices = [my_bus]
    if use_dist and dist.is_initialized() and world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, my_bus)
    n_physical = len(set(devices))
    if world > 1 and n_physical < world:
        shared_note = f"{world} ranks on {n_physical} GPU(s) (protocol check, not a scaling measurement)"
        not_measured = shared_note if not_measured is None else not_measured + "; " + shared_note

    result = {
        "metric": "RLVI samples/sec (fused E+M step)",
        "value": value, "unit": "samples/s", "n_gpus": n_physical, "n_ranks": world, "n_physical_gpus": n_physical,
        "steps": K, "warmup": W,
        "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": a.scaling if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic logits {B}x{C} per GPU, N={N} samples, "
                               "M-step (lagged pi) + E-step every step, HBM-cold rotation of "
                               f"{ROTATE} buffer pairs" +
                               ("" if not use_dist else
                                "; E-step sharded over the ranks (per-node totals through the peers' inboxes, "
                                "no collective call)" if estep_mode[0] == "sharded" else
                                "; residuals all-gathered, E-step replicated on every rank"),
                   "rows_per_gpu": B, "classes": C, "n_samples": N, "devices": devices,
                   **({"same_device": True} if (a.same_device and world > 1) else {}),
                   "launch": launch_mode.get("step", "eager"), "estep_dist": estep_mode[0],
                   "mstep_logits": "default form, no caller hint (what train_rlvi runs); the hinted form is "
                                   "parts.mstep_hinted_us",
                   **({"estep_dist_note": estep_note} if estep_note else {}),
                   **({"estep_dist_setup_s": round(setup_s, 3)} if setup_s is not None else {})},
    }
    if not_measured is not None:
        result["not_measured"] = not_measured
    if a.profile_only:
        if rank == 0:
            print(json.dumps(result))
        return

    # per-kernel legs (same buffers, same rotation): the HBM-bound M-step kernel and the
    # latency-bound E-step, each timed alone with HIP events on the launch stream
    KP = max(K, 50)                                  # calls per graph of a part leg (the driver runs --steps 20)

    def timed_part(fn, k=None, w=None, reps=3):
        """A part leg: a graph of k >= 50 calls, replayed `reps` times (each replay timed as the headline is);
        the median replay, per call, in ms."""
        k = KP if k is None else k
        w = max(W if w is None else w, 1)          # (at least one eager call before a capture: one-time set-up)
        ts = sorted(timed(fn, k, w, use_graph) / k for _ in range(reps))
        return ts[len(ts) // 2]

    ms_m = timed_part(mstep_only)
    ms_e = timed_part(estep_only)
    extra = {}
    if world == 1:
        from rlvi_amd import _lib as _lc
        Lc = _lc.load()
        # the same launch under the caller's hint that the block streams from HBM (a workspace option: the
        # timed hold of mstep.hip), on a workspace of its own
        ws_hint = ops.Workspace(dev, N, B)
        ops.hint_logits_from_hbm(ws_hint, True)

        def mstep_hinted(i, _ws):
            r = i % ROTATE
            ops.mstep_fwd_bwd(logits[r], labels, idx_local, weights, residuals, inv_scale=inv_scale,
                              grad=grads[r], ws=ws_hint, accumulate=True)
        extra["mstep_default_us"] = ms_m * 1e3
        extra["mstep_hinted_us"] = timed_part(mstep_hinted) * 1e3
        ops.mstep_reduce(ws=ws_hint)
        # a flat nontemporal 16-B/lane copy of the same bytes (logits block -> gradient block) in the same
        # rotation and graph: the plainest kernel that moves what the M-step moves -- this box's own ceiling
        def copy_only(i, _ws):
            r = i % ROTATE
            ops.stream_copy(grads[r], logits[r])
        extra["copy_same_bytes_us"] = timed_part(copy_only) * 1e3
        # single buffer pair: the block stays in the Infinity Cache between launches
        def mstep_warm(i, ws):
            ops.mstep_fwd_bwd(logits[0], labels, idx_local, weights, residuals, inv_scale=inv_scale,
                              grad=grads[0], ws=ws, accumulate=True)
        extra["mstep_warm_us"] = timed_part(mstep_warm) * 1e3
        ops.mstep_reduce(ws=ws)
        # ---- E-step beyond the warm case (estep_us: the same vector every call, the last call's trajectory is
        # the guess).  Drift: a different residual vector every call -- the bimodal NLLs scaled by 1.02^k plus
        # noise, k walking 0..4..0 -- as between the epochs of a training run: the guess is a neighbour's, not
        # this vector's.  Cold: no guess at all (workspace option cold_start), every call as the reference's loop
        # starts it (train_rlvi.py:29).  Each on a workspace of its own.
        rng_d = np.random.default_rng(11)
        r_base = residuals.detach().cpu().numpy().copy()
        drift_np = [(r_base * np.float32(1.02 ** k) + np.float32(0.01) * rng_d.random(N).astype(np.float32))
                    .astype(np.float32) for k in range(5)]
        walk = [0, 1, 2, 3, 4, 3, 2, 1]
        drift = [torch.from_numpy(v).to(dev) for v in drift_np]
        w_e = torch.ones(N, dtype=torch.float32, device=dev)
        it_e = torch.zeros(1, dtype=torch.int32, device=dev)
        ws_drift, ws_cold = ops.Workspace(dev, N, 0), ops.Workspace(dev, N, 0)
        ws_cold.set_option("cold_start", 1)

        def estep_drift(i, _ws):
            ops.estep_deep(drift[walk[i % len(walk)]], w_e, iters=it_e, ws=ws_drift)

        def estep_cold(i, _ws):
            ops.estep_deep(drift[walk[i % len(walk)]], w_e, iters=it_e, ws=ws_cold)
        extra["estep_drift_us"] = timed_part(estep_drift) * 1e3
        extra["estep_cold_us"] = timed_part(estep_cold) * 1e3
        extra["estep_cold_iters"] = int(it_e.item())
        # threshold + truncation over N samples (the epoch end once `overfit` is set; V4)
        thr_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        w_thr = weights.clone()

        def threshold_only(i, ws):
            _lc.check(Lc.rlvi_threshold_truncate_f32(ops._ptr(w_thr), N, 0.05, ops._ptr(thr_buf),
                                                     None, None, ws.ptr, ops._stream_ptr()),
                      "rlvi_threshold_truncate_f32")
        extra["threshold_us"] = timed_part(threshold_only) * 1e3
        extra["threshold_n"] = N
        # the criterion alone (false_negative_criterion, :41-49) on the E-step's pi as it is: the truncating call
        # above re-runs on its own output, whose thousands of exact zeros are not what an epoch end sees
        w_pi = weights.clone()

        def criterion_only(i, ws):
            _lc.check(Lc.rlvi_fn_threshold_f32(ops._ptr(w_pi), N, 0.05, ops._ptr(thr_buf), ws.ptr,
                                               ops._stream_ptr()), "rlvi_fn_threshold_f32")
        extra["criterion_us"] = timed_part(criterion_only) * 1e3
        # (the same vector every call: every guess from the previous call is right.  Without any guess:)
        def threshold_cold(i, _ws):
            _lc.check(Lc.rlvi_threshold_truncate_f32(ops._ptr(w_thr), N, 0.05, ops._ptr(thr_buf),
                                                     None, None, ws_cold.ptr, ops._stream_ptr()),
                      "rlvi_threshold_truncate_f32")

        def criterion_cold(i, _ws):
            _lc.check(Lc.rlvi_fn_threshold_f32(ops._ptr(w_pi), N, 0.05, ops._ptr(thr_buf), ws_cold.ptr,
                                               ops._stream_ptr()), "rlvi_fn_threshold_f32")
        extra["threshold_cold_us"] = timed_part(threshold_cold) * 1e3
        extra["criterion_cold_us"] = timed_part(criterion_cold) * 1e3
        # in-batch E+M (V2): NLL pass -> E-step on this batch -> weighted loss + gradient
        pi_b = torch.ones(B, dtype=torch.float32, device=dev)
        rows_b = torch.empty(B, dtype=torch.float32, device=dev)
        it_f = torch.zeros(1, dtype=torch.int32, device=dev)

        def fused_only(i, ws):
            r = i % ROTATE
            ops.fused_em(logits[r], labels, pi_b, ws=ws, out=out, grad=grads[r], rows=rows_b, iters=it_f)
        fe_ms = timed_part(fused_only)
        extra["fused_em_us"] = fe_ms * 1e3
        extra["fused_em_iters"] = int(it_f.item())
        # (one launch, the block resident in LDS between the passes, when the shape allows: fused_em.hip;
        #  V2 = 2*C*s + 16 bytes per sample -- logits in, gradient out, label, pi out, loss row out)
        extra["fused_em_frac"] = (B * (2 * C * 4 + 16) / (fe_ms * 1e-3)) / HBM_PEAK
        _lc.check(Lc.rlvi_tune_set(b"RLVI_FUSED_EM", 0), "tune")
        extra["fused_em_3launch_us"] = timed_part(fused_only) * 1e3
        Lc.rlvi_tune_unset(b"RLVI_FUSED_EM")
        # the M-step at 4x the rows (3 rotating pairs = 630 MB): the same kernel with the fixed
        # launch / ramp-up share of a 10-us launch amortised -- separates steady-state bandwidth
        # from ramp-up by measurement
        if (B, C) == (65536, 100):
            B4 = 4 * B
            gen = torch.Generator(device=dev)
            gen.manual_seed(7)
            big = [3.0 * torch.randn((B4, C), generator=gen, device=dev, dtype=torch.float32) for _ in range(3)]
            gbig = [torch.empty((B4, C), dtype=torch.float32, device=dev) for _ in range(3)]
            lab4 = labels.repeat(4)
            idx4 = torch.randperm(B4, generator=gen, device=dev)
            w4 = torch.rand(B4, generator=gen, device=dev)
            r4 = torch.zeros(B4, device=dev)
            ws4 = ops.Workspace(dev, B4, B4)

            def mstep_big(i, _ws):
                ops.mstep_fwd_bwd(big[i % 3], lab4, idx4, w4, r4, inv_scale=1.0 / B4, grad=gbig[i % 3],
                                  ws=ws4, accumulate=True)
            k4 = max(K // 4, 20)
            ms4 = timed_part(mstep_big, k4, 6)
            extra["mstep_4x_rows"] = B4
            extra["mstep_4x_us"] = ms4 * 1e3
            extra["mstep_4x_frac"] = (B4 * (2 * C * 4 + 24) / (ms4 * 1e-3)) / HBM_PEAK
            del big, gbig
        # ---- what each of BASELINE.json's five configs calls, at its own shape (gpu_us; the oracle's cpu_us is
        # added in the cpu_baseline leg below)
        cfg_cpu_jobs = {}
        extra["configs"] = config_legs(torch, np, dev, ops, timed_part, cfg_cpu_jobs, out)
        if not a.no_epoch_legs:
            extra["epoch"] = epoch_legs(torch, dev, a)
    bytes_per_sample = 2 * C * 4 + 24
    achieved = B * bytes_per_sample / (ms_m * 1e-3)
    # HBM bytes per launch from the PMC passes of tools/profile_bench.sh (FETCH_SIZE x2 on gfx950 +
    # WRITE_SIZE, separate rocprofv3 runs): a committed measurement of this kernel at this shape,
    # not something bench.py can collect live
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "mstep_traffic.json")
    if os.path.exists(tpath) and (B, C) == (65536, 100):
        traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
    result["roofline"] = {
        "bound": "hbm", "kernel": "rlvi::mstep_wave_kernel<float,4,4,7,16,true>",
        "form": "default: 16-wave workgroups, barrier behind the issue of the tile loads (no caller hint)",
        "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": achieved / HBM_PEAK, "traffic": traffic,
        "traffic_unit": "bytes per launch (profiles/mstep_traffic.json)",
        "algorithmic_bytes_per_launch": B * bytes_per_sample,
        "bytes_per_sample": bytes_per_sample, "us_per_launch": ms_m * 1e3,
        "step_frac": (B * bytes_per_sample / (ms_step * 1e-3)) / HBM_PEAK,
    }
    if "copy_same_bytes_us" in extra:
        # the flat copy moves 2 * C * 4 bytes per sample, the M-step 24 more (label, index, pi, residual)
        result["roofline"]["copy_same_bytes_us"] = extra["copy_same_bytes_us"]
        result["roofline"]["copy_frac_of_peak"] = (B * 2 * C * 4 / (extra["copy_same_bytes_us"] * 1e-6)) / HBM_PEAK
        result["roofline"]["frac_of_copy"] = extra["copy_same_bytes_us"] / (ms_m * 1e3)
        result["roofline"]["frac_hinted"] = (B * bytes_per_sample / (extra["mstep_hinted_us"] * 1e-6)) / HBM_PEAK
    result["parts"] = {"mstep_us": ms_m * 1e3, "estep_us": ms_e * 1e3,
                       "estep_iters": it_gpu, "estep_n": N,
                       "mstep_samples_per_s": B / (ms_m * 1e-3)}
    result["parts"].update(extra)
    result["parity"] = parity
    st = ws.status()
    for ev in status_log:
        st |= int(ev["status"])
    result["device_status"] = st          # OR over every leg of this run (0 = no device-side condition anywhere)
    if status_log:
        result["device_status_events"] = status_log

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import rlvi_oracle as O
        ncpu = os.cpu_count() or 1
        w0 = w_before.cpu().numpy()
        t_leg = time.perf_counter()

        def cpu_step(w_o, res_o):
            O.mstep(d0["logits"], d0["labels"], d0["idx"], w_o, res_o, scale_div=N)
            O.update_sample_weights(res_o, w_o)

        def median_time(fn, reps):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts))

        # the box exposes more logical CPUs than the job's share: pick the OpenMP thread count that is
        # fastest here -- one untimed + three timed steps per candidate, the MEDIAN decides (a single
        # timed step per candidate made the choice noise: 32 threads in round 1, 64 in round 2, a factor
        # of two in the reported baseline on unchanged oracle code)
        cal = {}
        for t in (8, 16, 24, 32, 48, 64, 96, 128):
            if t > ncpu:
                continue
            O.set_threads(t)
            w_o, res_o = w0.copy(), np.zeros(N, np.float32)
            cpu_step(w_o, res_o)
            cal[t] = median_time(lambda: cpu_step(w_o, res_o), 3)
        if not cal:
            cal[ncpu] = None
        best_t = min(cal, key=lambda t: cal[t] if cal[t] is not None else 0.0)
        O.set_threads(best_t)
        w_o, res_o = w0.copy(), np.zeros(N, np.float32)
        cpu_step(w_o, res_o)
        ts = []
        t_start = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            cpu_step(w_o, res_o)
            ts.append(time.perf_counter() - t0)
            el = time.perf_counter() - t_start
            if (el >= a.cpu_seconds and len(ts) >= 9) or len(ts) >= 5000:
                break
        med = float(np.median(ts))
        # the other variants of BASELINE.md 4.2 with the same oracle and thread count, median of 9 each
        res_c = np.zeros(N, np.float32)
        O.mstep(d0["logits"], d0["labels"], d0["idx"], w0.copy(), res_c, scale_div=N)

        def cpu_estep():
            O.update_sample_weights(res_c.copy(), np.ones(N, np.float32))

        w_pi = np.ones(N, np.float32)
        O.update_sample_weights(res_c.copy(), w_pi)

        def cpu_threshold():
            w_t = w_pi.copy()
            O.truncate(w_t, O.false_negative_criterion(w_t))

        def cpu_v2():
            l_b, _ = O.nll_rows(d0["logits"], d0["labels"])
            pi_b = np.ones(B, np.float32)
            O.update_sample_weights(l_b, pi_b)
            O.mstep(d0["logits"], d0["labels"], np.arange(B), pi_b, np.zeros(B, np.float32))

        parts_cpu = {}
        for name, fn in (("estep_alone_us", cpu_estep), ("threshold_truncate_us", cpu_threshold),
                         ("in_batch_em_v2_us", cpu_v2)):
            fn()
            parts_cpu[name] = median_time(fn, 9) * 1e6
        # the configs' CPU times: the same oracle on the same inputs -- the faster of one thread and the calibrated
        # thread count (a 4096 x 10 batch is over before 48 threads have started), median of 5
        if "configs" in result["parts"]:
            for (cfg, key), job in cfg_cpu_jobs.items():
                best, cores = None, None
                for t in sorted({1, best_t}):
                    O.set_threads(t)
                    job()
                    v = median_time(job, 5) * 1e6
                    if best is None or v < best:
                        best, cores = v, t
                result["parts"]["configs"][cfg][key] = round(best, 1)
                result["parts"]["configs"][cfg][key.replace("_us", "_cores")] = cores
            O.set_threads(best_t)
        result["cpu_baseline"] = {
            "value": B / med, "unit": "samples/s", "cores": best_t, "kind": "port",
            "sample": f"median of {len(ts)} full steps (M-step fwd+bwd + E-step) of the same {B}x{C} workload "
                      f"({el:.1f} s of CPU work); C oracle (oracle/rlvi_oracle.c) with OpenMP on {best_t} "
                      f"threads (fastest median of 3 among {sorted(cal)}; os.cpu_count()={ncpu})",
            "step_us": med * 1e6,
            "calibration_us": {str(t): (None if v is None else round(v * 1e6, 1)) for t, v in sorted(cal.items())},
            "parts": {k: round(v, 1) for k, v in parts_cpu.items()},
            "leg_seconds": round(time.perf_counter() - t_leg, 1),
        }
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        for p in peers_keep:
            p.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
