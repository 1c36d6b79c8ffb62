"""Epoch driver for RLVI only -- the caller side of the plug-in boundary.

Mirrors what deep-learning/main.py:run() does for `--method=rlvi` (`:196-350`): loaders that
yield (image, label, index), SGD with momentum and weight decay, the LR schedule (`:234-237`:
MultiplicativeLR with utils.get_lr_factor for MNIST, CosineAnnealingLR(T_max=200) otherwise,
stepped once per epoch `:322`), the evaluation of the initial model and its "epoch 0" log line
(`:257-262`), the per-epoch `train_rlvi` call (`:277-280`), the validation-based `overfit` detector
(`:283-288`), the selection mask `sample_weights > threshold` (`:343`) with the clean / corrupted
identification ratios (`utils.py:83-92`) and the TSV log line (`:347-350`).  The reference's own
driver cannot express large batches (hard-coded 32 / 128, `main.py:56,96`) and needs torchvision +
downloads; this one runs on synthetic data of the same shapes with any batch size, so
BASELINE.json's configs (MNIST-shaped, batch 4096, symmetric noise 0.5; CIFAR-10-shaped, ResNet18,
pairflip 0.45, batch sharded over the ranks) run end to end:

    python -m rlvi_amd.driver --dataset mnist --batch_size 4096 --n_train 54000 --n_epoch 4
    python -m torch.distributed.run --nproc-per-node 8 -m rlvi_amd.driver --dataset cifar10 --batch_size 32768

The models only produce logits and are stock torch.nn: a LeNet-shaped CNN (two 5x5 convs with 2x2
max-pools, 120-84-C head, as models/lenet.py:17-35) and a CIFAR-style ResNet18 (3x3 stem without
padding, four stages of two basic blocks at 64/128/256/512 channels, 1x1-conv shortcuts where the
shape changes, global average pool, linear head, as models/resnet.py:22-100).
"""
import argparse
import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim.lr_scheduler import CosineAnnealingLR, MultiplicativeLR

from . import dist as rdist
from .methods import train_rlvi

# per-dataset defaults of the reference (main.py:52-58, :92-97): epochs, batch, weight decay, lr
DATASETS = {
    "mnist": dict(input_channel=1, num_classes=10, image=28, n_epoch=100, batch_size=32, wd=1e-3, lr_init=0.01),
    "cifar10": dict(input_channel=3, num_classes=10, image=32, n_epoch=200, batch_size=128, wd=5e-4, lr_init=0.01),
}


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


def _conv_bn(cin, cout, k, stride, pad):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False), nn.BatchNorm2d(cout))


class _Residual(nn.Module):
    """Two 3x3 conv+BN with a ReLU between, added to the (possibly projected) input, ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.body = nn.Sequential(_conv_bn(cin, cout, 3, stride, 1), nn.ReLU(inplace=True),
                                  _conv_bn(cout, cout, 3, 1, 1))
        self.skip = nn.Identity() if (stride == 1 and cin == cout) else _conv_bn(cin, cout, 1, stride, 0)

    def forward(self, x):
        return F.relu(self.body(x) + self.skip(x))


class ResNet18(nn.Module):
    def __init__(self, input_channel=3, num_classes=10, widths=(64, 128, 256, 512), depth=(2, 2, 2, 2)):
        super().__init__()
        blocks, cin = [], widths[0]
        for stage, (w, n) in enumerate(zip(widths, depth)):
            for j in range(n):
                blocks.append(_Residual(cin, w, 2 if (j == 0 and stage > 0) else 1))
                cin = w
        self.stem = nn.Sequential(_conv_bn(input_channel, widths[0], 3, 1, 0), nn.ReLU(inplace=True))
        self.stages = nn.Sequential(*blocks)
        self.head = nn.Linear(cin, num_classes)

    def forward(self, x):
        out = self.stages(self.stem(x.float()))
        return self.head(torch.flatten(F.adaptive_avg_pool2d(out, 1), 1))


def get_lr_factor(epoch):
    """utils.get_lr_factor (deep-learning/utils.py:14-27): 1 for the first 20 epochs of 100, a linear
    ramp down to 0.01 until epoch 40, 0.01 afterwards (MultiplicativeLR multiplies it in every epoch)."""
    t = epoch / 100
    floor = 0.01
    if t <= 0.2:
        return 1.0
    if t <= 0.4:
        return 1.0 - (1.0 - floor) * (t - 0.2) / 0.2
    return floor


def synthetic_images(n, input_channel=1, num_classes=10, image=28, noise_rate=0.5, noise_type="symmetric",
                     seed=0):
    """Synthetic task of the dataset's shape: class prototypes + pixel noise.  Label noise as
    data_tools.py does it: `symmetric` re-draws `noise_rate` of the labels uniformly among the OTHER
    classes (:156-200), `pairflip` moves them to the next class (:100-152).
    Returns images [n,c,h,w] fp32, noisy labels, clean labels, noise_mask (True = clean)."""
    rng = np.random.default_rng(seed)
    protos = rng.standard_normal((num_classes, input_channel, image, image)).astype(np.float32)
    y = rng.integers(0, num_classes, n)
    x = protos[y] + 0.7 * rng.standard_normal((n, input_channel, image, image)).astype(np.float32)
    flip = rng.random(n) < noise_rate
    shift = rng.integers(1, num_classes, n) if noise_type == "symmetric" else np.ones(n, np.int64)
    y_noisy = np.where(flip, (y + shift) % num_classes, y)
    return (torch.from_numpy(x), torch.from_numpy(y_noisy.astype(np.int64)),
            torch.from_numpy(y.astype(np.int64)), ~flip)


def synthetic_digits(n, num_classes=10, noise_rate=0.5, seed=0, image=28):
    return synthetic_images(n, 1, num_classes, image, noise_rate, "symmetric", seed)


class IndexedLoader:
    """Yields (images, labels, indexes) like data_load.py:70 + DataLoader(shuffle, drop_last=False).

    rank / world: every batch is cut into `world` equal contiguous shards (a batch that does not
    divide is padded with its own first rows, as torch's DistributedSampler pads), so the ranks
    always run the same number of equally long batches."""

    def __init__(self, x, y, batch_size, shuffle, seed=0, rank=0, world=1):
        self.x, self.y, self.bs, self.shuffle = x, y, batch_size, shuffle
        self.rank, self.world = rank, world
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.y) + self.bs - 1) // self.bs

    def __iter__(self):
        n = len(self.y)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        for s in range(0, n, self.bs):
            ix = order[s:s + self.bs]
            if self.world > 1:
                per = (len(ix) + self.world - 1) // self.world
                if per * self.world != len(ix):
                    ix = torch.cat([ix, ix[:per * self.world - len(ix)]])
                ix = ix[self.rank * per:(self.rank + 1) * per]
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


LOG_HEADER = "epoch:\ttime_ep\ttau\tfix\tclean,%\tcorr,%\ttrain_acc\tval_acc\ttest_acc\n"


def run(n_train=16384, n_val=2048, n_test=2048, batch_size=1024, n_epoch=9, noise_rate=0.5,
        lr=0.1, momentum=0.9, wd=1e-3, seed=1, log_path=None, device="cuda", return_state=False,
        dataset="mnist", noise_type="symmetric", schedule=True, train_fn=None, evaluate_fn=None):
    """main.py:run() for RLVI on synthetic data.  Returns the list of per-epoch log dicts (the first
    one is the reference's "epoch 0" line: the initial model's test accuracy).

    train_fn / evaluate_fn: the per-epoch training function (default: this package's train_rlvi)
    and the evaluation function -- the tests pass a plain-torch restatement to check the driver's
    bookkeeping.  Under an initialised torch.distributed group the loaders shard every batch over
    the ranks and the model is wrapped in DistributedDataParallel."""
    cfg = DATASETS[dataset]
    device = torch.device(device)
    train_fn = train_fn or train_rlvi
    evaluate_fn = evaluate_fn or evaluate
    torch.manual_seed(seed)
    x, y, y_clean, noise_mask = synthetic_images(n_train + n_val + n_test, cfg["input_channel"],
                                                 cfg["num_classes"], cfg["image"], noise_rate, noise_type, seed)
    tr = slice(0, n_train)
    va = slice(n_train, n_train + n_val)
    te = slice(n_train + n_val, None)
    world = rdist.world_size()
    rank = torch.distributed.get_rank() if world > 1 else 0
    train_loader = IndexedLoader(x[tr], y[tr], batch_size, shuffle=True, seed=seed, rank=rank, world=world)
    val_loader = IndexedLoader(x[va], y[va], batch_size, shuffle=False)          # noisy, as main.py
    test_loader = IndexedLoader(x[te], y_clean[te], batch_size, shuffle=False)   # clean labels
    Model = LeNet if dataset == "mnist" else ResNet18
    model = Model(input_channel=cfg["input_channel"], num_classes=cfg["num_classes"]).to(device)
    net = model
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = DDP(model, device_ids=[device.index] if device.type == "cuda" and
                  torch.distributed.get_backend() == "nccl" else None)
    optimizer = torch.optim.SGD(net.parameters(), lr=lr, weight_decay=wd, momentum=momentum)
    scheduler = None
    if schedule:                                                   # main.py:234-237
        scheduler = (MultiplicativeLR(optimizer, get_lr_factor) if dataset == "mnist"
                     else CosineAnnealingLR(optimizer, T_max=200))

    sample_weights = torch.ones(n_train, device=device)            # main.py:250
    residuals = torch.zeros_like(sample_weights)                   # main.py:251
    overfit, threshold = False, 0                                  # main.py:252-253
    val_acc_old = val_acc_old_old = 0.0
    write = log_path and rank == 0
    # main.py:257-262: the initial model
    test_acc = rdist.rank0_value(evaluate_fn(test_loader, model, device))
    logs = [dict(epoch=0, time_ep=0.0, tau=0.0, fix=False, clean=100.0, corr=0.0, train_acc=0.0,
                 val_acc=0.0, test_acc=test_acc, lr=optimizer.param_groups[0]["lr"], kept=n_train)]
    if write:
        with open(log_path, "w") as f:
            f.write(LOG_HEADER)
            f.write(f"0:\t0\t0\t{False}\t100\t0\t0\t0\t{test_acc:8.4f}\n")
    for epoch in range(1, n_epoch):                                # main.py:265
        net.train()
        t0 = time.time()
        lr_now = optimizer.param_groups[0]["lr"]
        train_acc, threshold = train_fn(train_loader, net, optimizer, residuals,
                                        sample_weights, overfit, threshold)         # main.py:277-280
        # (under DDP every rank evaluates its own replica; BatchNorm's running statistics may differ by a
        #  rank's last batch, so the number that steers `overfit` is rank 0's on every rank)
        val_acc = rdist.rank0_value(evaluate_fn(val_loader, model, device))
        if not overfit:                                            # main.py:283-288
            if epoch > 2:
                overfit = val_acc < 0.5 * (val_acc_old + val_acc_old_old)
            val_acc_old_old, val_acc_old = val_acc_old, val_acc
        if scheduler is not None:
            scheduler.step()                                       # main.py:322
        time_ep = time.time() - t0
        test_acc = rdist.rank0_value(evaluate_fn(test_loader, model, device))
        mask = (sample_weights > threshold).cpu().numpy()          # main.py:343
        clean, corr = get_ratio_corrupted(mask, noise_mask[tr])
        rec = dict(epoch=epoch, time_ep=time_ep, tau=float(threshold), fix=bool(overfit),
                   clean=100 * clean, corr=100 * corr, train_acc=train_acc, val_acc=val_acc,
                   test_acc=test_acc, lr=lr_now, kept=int(mask.sum()))
        logs.append(rec)
        if write:
            with open(log_path, "a") as f:                         # main.py:347-350
                f.write(f"{epoch}:\t{time_ep:.2f}\t{float(threshold):.2f}\t{overfit}\t"
                        f"{100 * clean:.2f}\t{100 * corr:.2f}\t"
                        f"{train_acc:8.4f}\t{val_acc:8.4f}\t{test_acc:8.4f}\n")
    if return_state:
        return logs, sample_weights, noise_mask[tr]
    return logs


def main(argv=None):
    """Command line in the reference's vocabulary (main.py:20-36) plus the sizes it hard-codes."""
    ap = argparse.ArgumentParser(description="RLVI epoch driver on synthetic data (MI355X)")
    ap.add_argument("--result_dir", default="results/")
    ap.add_argument("--dataset", default="mnist", choices=sorted(DATASETS))
    ap.add_argument("--noise_rate", type=float, default=0.45)
    ap.add_argument("--noise_type", default="pairflip", choices=["pairflip", "symmetric"])
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--n_epoch", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=None, help="GLOBAL batch (sharded over the ranks)")
    ap.add_argument("--lr_init", type=float, default=None)
    ap.add_argument("--wd", type=float, default=None)
    ap.add_argument("--n_train", type=int, default=54000)
    ap.add_argument("--n_val", type=int, default=6000)
    ap.add_argument("--n_test", type=int, default=10000)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend under torchrun (nccl = RCCL)")
    a = ap.parse_args(argv)
    cfg = DATASETS[a.dataset]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        torch.distributed.init_process_group(a.backend)
    os.makedirs(os.path.join(a.result_dir, a.dataset, "rlvi"), exist_ok=True)
    log = os.path.join(a.result_dir, a.dataset, "rlvi",
                       f"{a.dataset}_rlvi_{a.noise_type}_{a.noise_rate}-s{a.seed}.txt")
    logs = run(n_train=a.n_train, n_val=a.n_val, n_test=a.n_test,
               batch_size=a.batch_size or cfg["batch_size"], n_epoch=a.n_epoch or cfg["n_epoch"],
               noise_rate=a.noise_rate, noise_type=a.noise_type, lr=a.lr_init or cfg["lr_init"],
               momentum=a.momentum, wd=a.wd if a.wd is not None else cfg["wd"], seed=a.seed, log_path=log,
               device=f"cuda:{local}", dataset=a.dataset)
    if world == 1 or torch.distributed.get_rank() == 0:
        for r in logs:
            print(r)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
