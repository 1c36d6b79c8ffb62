"""MI355X drop-in for the reference's `--method=rlvi` plug-in.

Same names, positional order, in-place mutation and return types as
deep-learning/methods/train_rlvi.py (the boundary of SURVEY.md 8(b)):

    train_rlvi(train_loader, model, optimizer, residuals, weights, overfit, threshold)
        -> (train_acc: float, threshold)                         reference :52-106
    update_sample_weights(residuals, weights, tol=1e-3, maxiter=40) -> None   reference :14-38
    false_negative_criterion(weights, alpha=0.05) -> 0-dim tensor            reference :41-49

Every tensor statement of the reference's hot loop runs in librlvi_gfx950.so; torch is used for
the model, the optimizer and device memory only.  `deep-learning/main.py` picks this module up
unchanged when `methods` resolves to this package.
"""
import torch

from .. import dist as rdist
from .. import ops

__all__ = ['train_rlvi']

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


@torch.no_grad()
def update_sample_weights(residuals, weights, tol=1e-3, maxiter=40):
    """Optimize Bernoulli probabilities in place (reference :14-38).

    residuals is min-shifted in place and weights overwritten, exactly as the reference does;
    one cooperative HIP launch, no host synchronisation."""
    ops.estep_deep(residuals.detach(), weights, tol=tol, maxiter=maxiter)


@torch.no_grad()
def false_negative_criterion(weights, alpha=0.05):
    """Threshold from the fixed probability (alpha) of type II error (reference :41-49)."""
    return ops.fn_threshold(weights, alpha=alpha)


def train_rlvi(train_loader, model, optimizer,
               residuals, weights, overfit, threshold):
    """Train one epoch with Bernoulli-probability-weighted SGD updates (reference :52-106).

    residuals, weights: caller-owned fp32 vectors of len(train_dataset) on DEVICE, mutated in
    place.  Returns (train_acc, threshold); threshold comes back unchanged while `overfit` is
    False and as a 0-dim device tensor once truncation has run (as in the reference).

    Under an initialised torch.distributed group (one process per GPU, the loader sharded over
    the ranks, the model wrapped in DistributedDataParallel) nothing changes for the caller:
    every rank streams its own rows -- DDP's gradient averaging over equal shards turns the local
    1/B into the single-device 1/B_global --, the residuals of the rows each rank visited are
    exchanged once before the E-step, every rank runs the E-step / threshold on the identical
    vector, and (train_acc, threshold) come back identical on every rank.

    Opt-in for loaders that give every rank the SAME samples every epoch
    (rlvi_amd.dist.set_owner_sharding): nothing is exchanged or gathered at all -- the E-step and the
    threshold run sharded over the ranks' own samples, the kernels passing their totals through the
    peers' inboxes -- and only the owned entries of residuals / weights are kept up to date on a rank."""
    train_total = 0
    owner = rdist.owner_sharding() if rdist.world_size() > 1 else None
    ws = owner[1] if owner is not None else ops.workspace(weights.device, weights.shape[0], 0)
    world = rdist.world_size()
    if world > 1:
        rdist.declare_device_sharing()      # (collective once per group, cached afterwards)
    visited, sizes = [], []
    # the batch loop's launcher, validated once per epoch (vectors, workspace, stream)
    mstep = ops.MStepLoop(weights, residuals.detach(), ws)

    for (images, labels, indexes) in train_loader:
        images = images.to(weights.device, non_blocking=True)
        labels = labels.to(weights.device, non_blocking=True)
        indexes = indexes.to(weights.device, non_blocking=True)

        logits = model(images)
        inv_scale = None                                       # 1 / B
        if world > 1:
            if owner is None:
                visited.append(indexes)
            sizes.append(int(labels.shape[0]))
            if rdist.ragged():                                 # unequal shards: world / B_global
                inv_scale = world / rdist.global_batch(labels.shape[0])
        # reference :85,:89-94 and the backward of :96 in ONE fused launch over the logits:
        # top-1, per-sample CE, residuals[indexes] = loss, weights[indexes] gather, weighted
        # mean and d(loss)/d(logits); the batch scalars accumulate on the device
        grad = mstep(logits, labels, indexes, inv_scale)
        train_total += 1

        optimizer.zero_grad()
        logits.backward(grad)          # == loss.backward() of the reference (:96)
        optimizer.step()

    if world > 1:
        if not rdist.ragged():
            rdist.check_equal_shards(sizes)
        if visited:
            rdist.exchange_residuals(residuals.detach(), torch.cat(visited))

    if owner is not None:
        # fixed ownership (rdist.set_owner_sharding): the same :99-103 on this rank's own samples, the
        # kernels exchanging their per-node / per-bin totals through the peers' inboxes -- no gather
        owned = owner[0]
        # (the kernels below wait for the other ranks' kernels, bounded at 100 x the spin bound: line the
        #  ranks up on the host first, so that a rank that is seconds behind -- I/O, an evaluation pass --
        #  is waited for here and not inside a spinning kernel)
        torch.distributed.barrier()
        r_own = residuals.detach()[owned].contiguous()
        w_own = weights[owned].contiguous()
        out = torch.empty(4, dtype=torch.float32, device=weights.device)
        ops.estep_sharded(r_own, w_own, weights.shape[0], batches=train_total, out=out, ws=ws)
        if overfit:
            threshold, _, _ = ops.threshold_truncate_sharded(w_own, weights.shape[0], threshold, ws=ws)
        residuals.detach()[owned] = r_own
        weights[owned] = w_own
        if not train_total:
            out = None
    else:
        # reference :99-103 plus the reduction of the accumulated top-1 percentages (:86-87,:105):
        # E-step, optional truncation and the epoch scalars in one cooperative launch (+ threshold)
        threshold, out = ops.epoch_end(residuals.detach(), weights, overfit=overfit,
                                       threshold=threshold, batches=train_total, ws=ws)
    if world > 1 and out is not None:
        rdist.mean_scalars(out[:2])                            # per-rank means over equal shards

    train_acc = float(out[1]) if train_total else float("nan")     # (the one host sync of the epoch)
    # out-of-range labels / indexes (the reference raises there), a cooperating launch whose
    # workgroups could not all be resident, a fixed point that did not converge: never silent
    ws.raise_on_status("train_rlvi")
    return train_acc, threshold
