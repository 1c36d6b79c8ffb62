"""Lab (CPU, stock torch only -- no kernel of this repo is involved): how stable is the small-loss training loop of
tests/test_gpu_parity.py::test_small_loss_baselines_train_loops against a perturbation of its arithmetic?

Round 3 recorded ONE collapse of that loop (test accuracy 9 %, gpurun_out/t_small.txt) inside a full `-m gpu`
process at lr 0.1 and none in 40 repetitions in a fresh process (gpurun_out/dbg_usdnl.txt).  Those 40 repetitions
re-seed torch identically, so they are 40 copies of ONE trajectory (they differ by the order of a few atomic adds
in the convolutions' backward), not 40 samples.  What differs between the two processes and reaches the arithmetic
is the convolution algorithm MIOpen picks (its in-process find cache after the suite's earlier LeNet / ResNet
calls): a relative perturbation of 1e-7 ... 1e-6 on the conv outputs.  This script applies perturbations of that
size to the same loop as stock torch ops on the CPU: the data, the loader order and the model initialisation are
those of the test; only a relative noise of `--eps` on the logits, drawn with a per-run generator, varies.

    python tools/lab/usdnl_lr_stability.py --runs 24 --lr 0.1
    python tools/lab/usdnl_lr_stability.py --runs 24 --lr 0.05
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlvi_amd import driver  # noqa: E402  (LeNet, the synthetic digits and the loader: stock torch.nn / numpy)


def one_run(lr, eps, run, n_epoch=12):
    torch.manual_seed(0)
    xa, ya_noisy, ya_clean, _ = driver.synthetic_digits(5120, noise_rate=0.4, seed=3)
    x, y_noisy = xa[:4096], ya_noisy[:4096]
    xt, yt = xa[4096:], ya_clean[4096:]
    rate = np.ones(n_epoch) * 0.4
    rate[:4] = np.linspace(0, 0.4, 4)
    model = driver.LeNet()
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9)
    loader = driver.IndexedLoader(x, y_noisy, 256, shuffle=True, seed=1)
    gen = torch.Generator().manual_seed(1000 + run)
    worst = 0.0
    for epoch in range(n_epoch):
        model.train()
        for data, labels, _ in loader:
            logits = model(data)
            if eps > 0:
                logits = logits * (1.0 + eps * torch.randn(logits.shape, generator=gen))
            k = int((1 - rate[epoch]) * logits.shape[0])
            ce = F.cross_entropy(logits, labels, reduction='none')
            keep = torch.argsort(ce.detach(), stable=True)[:k]
            loss = ce[keep].mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            worst = max(worst, float(loss))
    model.eval()
    with torch.no_grad():
        acc = 100.0 * float((model(xt).argmax(1) == yt).sum()) / yt.numel()
    return acc, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=24)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--eps", type=float, default=1e-6)
    a = ap.parse_args()
    torch.set_num_threads(8)
    bad = 0
    for r in range(a.runs):
        acc, worst = one_run(a.lr, a.eps if r > 0 else 0.0, r)
        bad += acc < 85.0
        print(f"lr {a.lr} eps {a.eps if r > 0 else 0.0:g} run {r}: test {acc:.2f} %  largest batch loss {worst:.3f}",
              flush=True)
    print(f"collapsed (test accuracy < 85 %): {bad} of {a.runs} at lr {a.lr}")


if __name__ == "__main__":
    main()
