"""Epoch driver for RLVI only -- the caller side of the plug-in boundary.

Mirrors what deep-learning/main.py:run() does for `--method=rlvi` (`:196-350`): loaders that
yield (image, label, index), SGD with momentum, the per-epoch `train_rlvi` call, the
validation-based `overfit` detector (`:283-288`), the selection mask `sample_weights > threshold`
(`:343`) with the clean / corrupted identification ratios (`utils.py:83-92`) and the TSV log line
(`:347-350`).  The reference's own driver cannot express large batches (hard-coded 32 / 128,
`main.py:56,96`) and needs torchvision + downloads; this one runs on synthetic data with any batch
size, so BASELINE.json's configs (MNIST-shaped, batch 4096, symmetric noise 0.5) run end to end.

The model is stock PyTorch (it only produces logits): a LeNet-shaped CNN written with torch.nn,
as deep-learning/models/lenet.py:17-35 does (5x5 convs, 2x2 max-pools, 120-84-C head).
"""
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .methods import train_rlvi


class LeNet(nn.Module):
    def __init__(self, input_channel=1, num_classes=10):
        super().__init__()
        self.conv1 = nn.Conv2d(input_channel, 6, kernel_size=5)
        self.conv2 = nn.Conv2d(6, 16, kernel_size=5)
        self.fc1 = nn.Linear(16 * 4 * 4, 120)
        self.fc2 = nn.Linear(120, 84)
        self.fc3 = nn.Linear(84, num_classes)

    def forward(self, x):
        out = F.max_pool2d(F.relu(self.conv1(x)), 2)
        out = F.max_pool2d(F.relu(self.conv2(out)), 2)
        out = out.view(out.shape[0], -1)
        return self.fc3(F.relu(self.fc2(F.relu(self.fc1(out)))))


def synthetic_digits(n, num_classes=10, noise_rate=0.5, seed=0, image=28):
    """MNIST-shaped synthetic task: class prototypes + pixel noise; `noise_rate` of the labels are
    re-drawn uniformly among the OTHER classes (symmetric noise, data_tools.py:156-200 semantics).
    Returns images [n,1,28,28] fp32, noisy labels, clean labels, noise_mask (True = clean)."""
    rng = np.random.default_rng(seed)
    protos = rng.standard_normal((num_classes, image, image)).astype(np.float32)
    y = rng.integers(0, num_classes, n)
    x = protos[y] + 0.7 * rng.standard_normal((n, image, image)).astype(np.float32)
    flip = rng.random(n) < noise_rate
    shift = rng.integers(1, num_classes, n)
    y_noisy = np.where(flip, (y + shift) % num_classes, y)
    return (torch.from_numpy(x[:, None]), torch.from_numpy(y_noisy.astype(np.int64)),
            torch.from_numpy(y.astype(np.int64)), ~flip)


class IndexedLoader:
    """Yields (images, labels, indexes) like data_load.py:70 + DataLoader(shuffle, drop_last=False)."""

    def __init__(self, x, y, batch_size, shuffle, seed=0):
        self.x, self.y, self.bs, self.shuffle = x, y, batch_size, shuffle
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.y) + self.bs - 1) // self.bs

    def __iter__(self):
        n = len(self.y)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        for s in range(0, n, self.bs):
            ix = order[s:s + self.bs]
            yield self.x[ix], self.y[ix], ix


@torch.no_grad()
def evaluate(loader, model, device):
    """utils.evaluate (deep-learning/utils.py:48-62): top-1 accuracy in percent."""
    from . import ops
    model.eval()
    hits = torch.zeros((), device=device)
    total = 0
    for images, labels, _ in loader:
        out = ops.evaluate_batch(model(images.to(device)), labels.to(device))   # one pass: CE + top-1
        hits += out[3]
        total += labels.numel()
    return 100.0 * float(hits) / total


def get_ratio_corrupted(mask, noise_mask):
    """utils.get_ratio_corrupted (deep-learning/utils.py:83-92)."""
    mask = np.asarray(mask, bool)
    clean_found = np.logical_and(mask, noise_mask).sum() / max(1, noise_mask.sum())
    corr_found = np.logical_and(~mask, ~noise_mask).sum() / max(1, (~noise_mask).sum())
    return clean_found, corr_found


def run(n_train=16384, n_val=2048, n_test=2048, batch_size=1024, n_epoch=9, noise_rate=0.5,
        lr=0.1, momentum=0.9, wd=1e-3, seed=1, log_path=None, device="cuda", return_state=False):
    """main.py:run() for RLVI on synthetic data.  Returns the list of per-epoch log dicts."""
    device = torch.device(device)
    torch.manual_seed(seed)
    x, y, y_clean, noise_mask = synthetic_digits(n_train + n_val + n_test, noise_rate=noise_rate, seed=seed)
    tr = slice(0, n_train)
    va = slice(n_train, n_train + n_val)
    te = slice(n_train + n_val, None)
    train_loader = IndexedLoader(x[tr], y[tr], batch_size, shuffle=True, seed=seed)
    val_loader = IndexedLoader(x[va], y[va], batch_size, shuffle=False)          # noisy, as main.py
    test_loader = IndexedLoader(x[te], y_clean[te], batch_size, shuffle=False)   # clean labels
    model = LeNet().to(device)
    optimizer = torch.optim.SGD(model.parameters(), lr=lr, weight_decay=wd, momentum=momentum)

    sample_weights = torch.ones(n_train, device=device)            # main.py:250
    residuals = torch.zeros_like(sample_weights)                   # main.py:251
    overfit, threshold = False, 0                                  # main.py:252-253
    val_acc_old = val_acc_old_old = 0.0
    logs = []
    if log_path:
        with open(log_path, "w") as f:
            f.write("epoch:\ttime_ep\ttau\tfix\tclean,%\tcorr,%\ttrain_acc\tval_acc\ttest_acc\n")
    for epoch in range(1, n_epoch):                                # main.py:265
        model.train()
        t0 = time.time()
        train_acc, threshold = train_rlvi(train_loader, model, optimizer, residuals,
                                          sample_weights, overfit, threshold)       # main.py:277-280
        val_acc = evaluate(val_loader, model, device)
        if not overfit:                                            # main.py:283-288
            if epoch > 2:
                overfit = val_acc < 0.5 * (val_acc_old + val_acc_old_old)
            val_acc_old_old, val_acc_old = val_acc_old, val_acc
        time_ep = time.time() - t0
        test_acc = evaluate(test_loader, model, device)
        mask = (sample_weights > threshold).cpu().numpy()          # main.py:343
        clean, corr = get_ratio_corrupted(mask, noise_mask[tr])
        rec = dict(epoch=epoch, time_ep=time_ep, tau=float(threshold), fix=bool(overfit),
                   clean=100 * clean, corr=100 * corr, train_acc=train_acc, val_acc=val_acc,
                   test_acc=test_acc)
        logs.append(rec)
        if log_path:
            with open(log_path, "a") as f:                         # main.py:347-350
                f.write(f"{epoch}:\t{time_ep:.2f}\t{float(threshold):.2f}\t{overfit}\t"
                        f"{100 * clean:.2f}\t{100 * corr:.2f}\t"
                        f"{train_acc:8.4f}\t{val_acc:8.4f}\t{test_acc:8.4f}\n")
    if return_state:
        return logs, sample_weights, noise_mask[tr]
    return logs


if __name__ == "__main__":
    for r in run(log_path=None):
        print(r)
