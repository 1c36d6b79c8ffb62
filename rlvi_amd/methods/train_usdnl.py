"""MI355X mirror of the small-loss baseline deep-learning/methods/train_usdnl.py.

Same names and signatures: loss_fn(logits, labels, forget_rate) -> loss (reference :16-27),
train_usdnl(train_loader, epoch, model, optimizer, rate_schedule) -> train_acc (:30-56).
The reference moves the per-sample losses to the host, argsorts them there and averages the kept
ones; here the per-sample CE, the selection (rlvi_select_smallest_f32) and the masked mean +
gradient (the streaming M-step kernel with a 0/1 weight vector) are three launches on the device.
"""
import torch

from .. import ops

__all__ = ['train_usdnl']

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def loss_fn(logits, labels, forget_rate):
    B = logits.shape[0]
    remember_rate = 1 - forget_rate
    num_remember = int(remember_rate * B)                    # (:22-23)
    if num_remember <= 0:                                    # torch.mean of an empty selection
        return logits.sum() * float('nan')
    labels = labels.long()
    with torch.no_grad():
        loss_pick = ops.per_sample_ce(logits.detach(), labels)            # (:17)
        mask = ops.select_smallest(loss_pick, num_remember)               # (:19-24)
    return ops.selected_cross_entropy(logits, labels, mask, 1.0 / num_remember)   # (:26)


def train_usdnl(train_loader, epoch, model, optimizer, rate_schedule):
    model.train()
    hits = torch.zeros((), device=DEVICE)
    train_total = 0
    for (data, labels, indexes) in train_loader:
        data = data.to(DEVICE)
        labels = labels.to(DEVICE)
        logits = model(data)
        with torch.no_grad():                                # accuracy(...)[0]: top-1 % (:42)
            hits += ops.evaluate_batch(logits.detach(), labels)[1]
        train_total += 1
        loss = loss_fn(logits, labels, rate_schedule[epoch])
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    return float(hits) / float(train_total)
