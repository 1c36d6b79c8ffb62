"""MI355X mirror of the small-loss baseline deep-learning/methods/train_coteaching.py
(SURVEY 8(f)-4): loss_coteaching(y_1, y_2, t, forget_rate, ind) -> (loss_1, loss_2) (reference
:17-35) and train_coteaching(...) -> train_acc1 (:39-76).

Each network is trained on the samples the OTHER one finds easy.  The reference argsorts both loss
vectors on the host and gathers sub-batches; here each model's rows get the other model's 0/1
selection as weights in the streaming M-step kernel.  Note the reference's normalisation:
F.cross_entropy(...) already averages over the kept rows and the result is divided by
num_remember once more (:32-35), i.e. loss = sum / num_remember^2 -- reproduced as is.
"""
import torch

from .. import ops

__all__ = ['train_coteaching']

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def loss_coteaching(y_1, y_2, t, forget_rate, ind):
    B = y_1.shape[0]
    remember_rate = 1 - forget_rate
    num_remember = int(remember_rate * B)                    # (:27-28)
    if num_remember <= 0:
        return y_1.sum() * float('nan'), y_2.sum() * float('nan')
    t = t.long()
    with torch.no_grad():
        loss_1 = ops.per_sample_ce(y_1.detach(), t)                       # (:18)
        loss_2 = ops.per_sample_ce(y_2.detach(), t)                       # (:22)
        keep_1 = ops.select_smallest(loss_1, num_remember)                # (:19,:30)
        keep_2 = ops.select_smallest(loss_2, num_remember)                # (:23,:31)
    inv = 1.0 / (float(num_remember) * float(num_remember))
    loss_1_update = ops.selected_cross_entropy(y_1, t, keep_2, inv)       # exchange (:33)
    loss_2_update = ops.selected_cross_entropy(y_2, t, keep_1, inv)       # (:34)
    return loss_1_update, loss_2_update


def train_coteaching(train_loader, epoch, model1, optimizer1, model2, optimizer2, rate_schedule):
    hits = torch.zeros((), device=DEVICE)
    train_total = 0
    for (images, labels, indexes) in train_loader:
        ind = indexes.cpu().numpy().transpose()
        images = images.to(DEVICE)
        labels = labels.to(DEVICE)
        logits1 = model1(images)
        logits2 = model2(images)
        with torch.no_grad():
            hits += ops.evaluate_batch(logits1.detach(), labels)[1]
        train_total += 1
        loss_1, loss_2 = loss_coteaching(logits1, logits2, labels, rate_schedule[epoch], ind)
        optimizer1.zero_grad()
        loss_1.backward()
        optimizer1.step()
        optimizer2.zero_grad()
        loss_2.backward()
        optimizer2.step()
    return float(hits) / float(train_total)
