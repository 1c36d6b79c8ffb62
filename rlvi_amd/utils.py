"""MI355X mirror of the metric helpers of deep-learning/utils.py that sit on the hot path's boundary.

    accuracy(logit, target, topk=(1,))        reference utils.py:65-79

`train_rlvi` itself keeps only precision@1 (train_rlvi.py:85) and gets it from the streaming M-step kernel; this is
the stand-alone helper for callers of the reference's function (e.g. topk=(1, 5)).
"""
from . import ops


def accuracy(logit, target, topk=(1,)):
    """Computes the precision@k for the specified values of k (reference utils.py:65-79): a list of one-element fp32
    device tensors, 100 * (rows whose label is among the k largest) / batch size.  One pass over the logits on the GPU
    (rlvi_topk_hits_*), no softmax, no sort; RuntimeError when max(topk) exceeds the number of classes, as the
    reference's torch.topk raises."""
    batch_size = target.size(0)
    hits = ops.topk_hits(logit, target.view(-1), topk)
    return [hits[i:i + 1].float().mul_(100.0 / batch_size) for i in range(len(topk))]
