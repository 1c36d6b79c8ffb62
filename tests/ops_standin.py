"""Test-only stand-in for rlvi_amd.ops backed by the CPU oracle, so that the HOST logic of the
plug-in (rlvi_amd/methods/train_rlvi.py: loop, distributed exchange, status handling) can run on a
box without a GPU.  Never imported by the product; the GPU tests use the real ops."""
import numpy as np
import torch

from oracle import rlvi_oracle as O


class FakeWorkspace:
    def __init__(self, status=0):
        self._status = status
        self.acc = []            # per-batch (loss, top-1 %)

    def status(self):
        return self._status

    def clear_status(self):
        self._status = 0

    def raise_on_status(self, what, mask=7):
        from rlvi_amd import _lib
        st = self.status()
        if st & mask:
            self.clear_status()
            raise _lib.RlviError(f"{what}: device status {st}: {_lib.status_message(st & mask)}")
        return st


class StandIn:
    """Same call signatures as the functions train_rlvi uses."""

    def __init__(self, status=0):
        self.ws = FakeWorkspace(status)

    def workspace(self, device, n=0, b=0):
        return self.ws

    def mstep_fwd_bwd(self, logits, labels, idx, weights, residuals, inv_scale=None, want_grad=True,
                      out=None, grad=None, ws=None, accumulate=False):
        B = logits.shape[0]
        w = weights.numpy()                      # shares memory with the torch tensor (CPU)
        r = residuals.numpy()
        scale_div = None if inv_scale is None else int(round(1.0 / float(inv_scale)))
        ref = O.mstep(logits.detach().numpy().astype(np.float32), labels.numpy().astype(np.int64),
                      idx.numpy().astype(np.int64), w, r, scale_div=scale_div)
        ws.acc.append((float(ref["loss"]), float(ref["prec1"])))
        return None, torch.from_numpy(ref["grad"].astype(np.float32))

    def MStepLoop(self, weights, residuals, ws=None):
        """The per-epoch launcher of the batch loop (rlvi_amd.ops.MStepLoop): same call, oracle arithmetic."""
        outer = self

        def call(logits, labels, idx, inv_scale=None):
            return outer.mstep_fwd_bwd(logits, labels, idx, weights, residuals, inv_scale=inv_scale,
                                       accumulate=True, ws=ws or outer.ws)[1]
        return call

    def epoch_end(self, residuals, weights, overfit=False, threshold=0, batches=0, tol=1e-3, maxiter=40,
                  alpha=0.05, out=None, iters=None, ws=None):
        O.update_sample_weights(residuals.numpy(), weights.numpy(), tol=tol, maxiter=maxiter)
        if overfit:
            thr = np.float32(max(np.float32(threshold), O.false_negative_criterion(weights.numpy(), alpha)))
            O.truncate(weights.numpy(), thr)
            threshold = torch.tensor(thr)
        o = None
        if batches > 0:
            a = np.array(ws.acc, np.float64)
            ws.acc.clear()
            o = torch.tensor([a[:, 0].mean(), a[:, 1].mean(), 0.0, 0.0], dtype=torch.float32)
        return threshold, o
