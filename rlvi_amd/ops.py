"""torch-facing wrappers of the C ABI (raw device pointers + current HIP stream).

PyTorch is plumbing here: device memory, streams, autograd glue.  All arithmetic of the
hot path happens in librlvi_gfx950.so; there is no eager/CPU fallback -- a missing library
or a non-GPU tensor raises.
"""
import ctypes

import torch

from . import _lib

_workspaces = {}


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# the current stream's handle without building a torch.cuda.Stream object (MStepLoop asks once per batch)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
    lambda dev: torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.RlviError(
                "rlvi_amd runs only on an MI355X HIP device: got a CPU tensor "
                "(there is no CPU fallback; the CPU restatement under oracle/ is test-only)")


class Workspace:
    """Caller-owned scratch + control words (include/rlvi_hip.h, 'Conventions')."""

    def __init__(self, device, max_n, max_b):
        L = _lib.load()
        self.max_n, self.max_b = int(max_n), int(max_b)
        self.nbytes = int(L.rlvi_workspace_bytes(self.max_n, self.max_b))
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        assert self.buf.data_ptr() % 256 == 0
        # a sharded call that timed out leaves the ranks' round counters / warm-start state out of step:
        # further sharded calls are refused until rlvi_amd.dist.setup_peers(ws) has run again (all ranks)
        self.peers_stale = False
        _lib.check(L.rlvi_workspace_init(_ptr(self.buf), self.nbytes, _stream_ptr()),
                   "rlvi_workspace_init")

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr())

    def status(self):
        s = ctypes.c_int32(0)
        _lib.check(_lib.load().rlvi_workspace_status(self.ptr, ctypes.byref(s), _stream_ptr()),
                   "rlvi_workspace_status")
        return s.value

    def clear_status(self):
        _lib.check(_lib.load().rlvi_workspace_clear_status(self.ptr, _stream_ptr()),
                   "rlvi_workspace_clear_status")

    def set_option(self, name, value):
        """Per-workspace launch option (include/rlvi_hip.h, rlvi_workspace_set_option): "logits_from_hbm",
        "cold_start".  Host side, from the next launch on this workspace on."""
        _lib.check(_lib.load().rlvi_workspace_set_option(self.ptr, name.encode(), int(value)),
                   f"rlvi_workspace_set_option({name})")

    def reset_warm(self):
        """Forget the guesses the last E-step / threshold left for the next one."""
        _lib.check(_lib.load().rlvi_workspace_reset_warm(self.ptr, _stream_ptr()), "rlvi_workspace_reset_warm")

    def region(self, name):
        """(offset, bytes) of a named region of the layout ("records", "records_out", "warm", "scratch")."""
        n = ctypes.c_size_t(0)
        off = int(_lib.load().rlvi_workspace_region(name.encode(), ctypes.byref(n)))
        if off == ctypes.c_size_t(-1).value:
            raise ValueError(f"unknown workspace region {name!r}")
        return off, int(n.value)

    def pending_records(self):
        """True if accumulate-mode M-step records are waiting for an epoch_end / mstep_reduce (synchronises)."""
        off, n = self.region("records")
        return bool(self.buf[off:off + n].view(torch.float64).ne(0).any().item())

    def raise_on_status(self, what, mask=_lib.ST_RANGE | _lib.ST_TIMEOUT | _lib.ST_NOCONV):
        """Read the sticky device status (one 4-byte copy + stream sync) and raise RlviError on any
        flag of `mask`; the flag is cleared first so that the caller can recover and go on."""
        st = self.status()
        if st & mask:
            self.clear_status()
            if st & _lib.ST_TIMEOUT:
                self.peers_stale = True
            raise _lib.RlviError(f"{what}: device status {st}: {_lib.status_message(st & mask)}")
        return st


def debug_scratch_offset():
    """Byte offset of the workspace scratch area (where RLVI_TJ_DEBUG=1 leaves its stamps): the
    fixed regions come first, so it is the size of a workspace for empty vectors minus its pad."""
    return int(_lib.load().rlvi_workspace_bytes(0, 0)) - 256


def workspace(device, n=0, b=0):
    """Per (device, stream) workspace, grown on demand."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.max_n < n or ws.max_b < b:
        ws = Workspace(device, max(n, ws.max_n if ws else 0, 1 << 16),
                       max(b, ws.max_b if ws else 0, 1 << 16))
        _workspaces[key] = ws
    return ws


def mstep_fwd_bwd(logits, labels, idx, weights, residuals, inv_scale=None, want_grad=True,
                  out=None, grad=None, ws=None, accumulate=False):
    """One mini-batch of the M-step (train_rlvi.py:85-96 without model/optimizer).

    Scatters the per-sample NLL into `residuals[idx]`, gathers the lagged pi from
    `weights[idx]`, returns (out, grad) with out = fp32[4] device tensor
    {weighted mean loss, top-1 %, sum pi*l, hits} and grad = dL/dlogits (or None).
    accumulate=True: ONE launch, no scalars now (out is None): the per-batch sums pile up in the
    workspace until `epoch_end` / `mstep_reduce` collects them.
    """
    L = _lib.load()
    _require_gpu(logits, labels, idx, weights, residuals)
    if logits.dim() != 2:
        raise ValueError("logits must be [B, C]")
    B, C = logits.shape
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    if labels.dtype != torch.int64 or not labels.is_contiguous():
        labels = labels.to(torch.int64).contiguous()
    if idx is not None and (idx.dtype != torch.int64 or not idx.is_contiguous()):
        idx = idx.to(torch.int64).contiguous()
    if weights.dtype != torch.float32 or not weights.is_contiguous():
        raise ValueError("weights must be a contiguous fp32 vector (it is read in place)")
    if residuals is not None and (residuals.dtype != torch.float32 or not residuals.is_contiguous()):
        raise ValueError("residuals must be a contiguous fp32 vector (it is written in place)")
    N = weights.shape[0]
    if accumulate:
        out = None
    elif out is None:
        out = torch.empty(4, dtype=torch.float32, device=logits.device)
    if want_grad and grad is None:
        grad = torch.empty((B, C), dtype=logits.dtype, device=logits.device)
    if not want_grad:
        grad = None
    ws = ws or workspace(logits.device, N, B)
    fn = L.rlvi_mstep_fwd_bwd_f32 if logits.dtype == torch.float32 else L.rlvi_mstep_fwd_bwd_bf16
    rc = fn(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(idx), _ptr(weights),
            _ptr(residuals), N, B, C, float(inv_scale if inv_scale is not None else 1.0 / B),
            _ptr(grad), grad.stride(0) if grad is not None else 0, _ptr(out), ws.ptr,
            _stream_ptr())
    _lib.check(rc, "rlvi_mstep_fwd_bwd")
    return out, grad


def hint_logits_from_hbm(ws, flag=True):
    """Tell the M-step launcher where the logits of the calls ON THIS WORKSPACE come from.

    True: the [B, C] block streams from HBM -- it is larger than the Infinity Cache, or one of many blocks
    touched in rotation.  A one-tile-per-wave launch of 12 MB and more then holds its gradient stores until its
    reads have had their time at the HBM read rate (reads first, then writes, chip-wide: mstep.hip).  False (the
    default): nothing is assumed -- a chip-filling launch separates its reads from its writes per CU with a
    barrier behind the issue of its loads; the timed hold would cost time on a block that the model's last
    layer has just written and the cache serves (9.4 -> 10.3 us), which is what train_rlvi sees.
    Per workspace (round 3 had a process-wide knob): two loops on two streams do not see each other's hint."""
    ws.set_option("logits_from_hbm", 1 if flag else 0)


class MStepLoop:
    """mstep_fwd_bwd(..., accumulate=True) for a training loop, validated ONCE: the per-epoch vectors
    (weights, residuals), the workspace, the stream and the ctypes entry are fixed when the object is made
    (train_rlvi makes one per epoch), so a batch costs a handful of data_ptr() reads, one cached gradient
    buffer per batch shape and the foreign call -- the generic wrapper's ~10 checks, dict look-ups,
    c_void_p objects and current-stream query are what a 4-microsecond kernel at 4096 x 10 was waiting for.
    A batch that does not look like the validated form (strided / non-fp32-bf16 logits, labels or indexes
    that are not contiguous int64 device tensors) takes the generic, fully checked path.

    The returned gradient buffer is REUSED by the next batch of the same shape: consume it
    (`logits.backward(grad)`) before the next call, as train_rlvi does."""

    def __init__(self, weights, residuals, ws=None):
        L = _lib.load()
        _require_gpu(weights, residuals)
        for t in (weights, residuals):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
                raise ValueError("weights / residuals must be contiguous 1-D fp32 tensors")
        if residuals.shape != weights.shape:
            raise ValueError("residuals and weights differ in length")
        self.weights, self.residuals = weights, residuals
        self.N = weights.shape[0]
        self.ws = ws or workspace(weights.device, self.N, 0)
        self._w, self._r, self._wsp = weights.data_ptr(), residuals.data_ptr(), self.ws.buf.data_ptr()
        self._stream = torch.cuda.current_stream(weights.device).cuda_stream
        self._f32, self._bf16 = L.rlvi_mstep_fwd_bwd_f32, L.rlvi_mstep_fwd_bwd_bf16
        self._grads = {}
        self._dev = weights.device
        self._devidx = weights.device.index if weights.device.index is not None else torch.cuda.current_device()

    def __call__(self, logits, labels, idx, inv_scale=None):
        dt = logits.dtype
        # the validated form, on the device and the stream the loop was made on; anything else -- strided or
        # other-typed logits, no index vector, a tensor on another device, a caller that has switched streams
        # inside the epoch -- takes the generic, fully checked wrapper (which launches on the CURRENT stream)
        if not (idx is not None and (dt is torch.float32 or dt is torch.bfloat16) and logits.dim() == 2
                and logits.is_contiguous() and labels.dtype is torch.int64 and idx.dtype is torch.int64
                and labels.is_contiguous() and idx.is_contiguous()
                and logits.device == self._dev and labels.is_cuda and idx.is_cuda
                and _raw_stream(self._devidx) == self._stream):
            _require_gpu(logits, labels, idx)
            _, grad = mstep_fwd_bwd(logits.detach(), labels, idx, self.weights, self.residuals,
                                    inv_scale=inv_scale, accumulate=True, ws=self.ws)
            return grad
        B, C = logits.shape
        key = (B, C, dt)
        grad = self._grads.get(key)
        if grad is None:
            grad = self._grads[key] = torch.empty((B, C), dtype=dt, device=self._dev)
        rc = (self._f32 if dt is torch.float32 else self._bf16)(
            logits.data_ptr(), C, labels.data_ptr(), idx.data_ptr(), self._w, self._r, self.N, B, C,
            float(inv_scale) if inv_scale is not None else 1.0 / B, grad.data_ptr(), C, None, self._wsp,
            self._stream)
        if rc:
            _lib.check(rc, "rlvi_mstep_fwd_bwd")
        return grad


def estep_deep(residuals, weights, tol=1e-3, maxiter=40, iters=None, trace=None, ws=None):
    """update_sample_weights (train_rlvi.py:14-38), in place on both vectors."""
    L = _lib.load()
    _require_gpu(residuals, weights)
    for t in (residuals, weights):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
            raise ValueError("residuals / weights must be contiguous 1-D fp32 tensors")
    if residuals.shape != weights.shape:
        raise ValueError("residuals and weights differ in length")
    N = weights.shape[0]
    ws = ws or workspace(weights.device, N, 0)
    rc = L.rlvi_estep_deep_f32(_ptr(residuals), _ptr(weights), N, float(tol), int(maxiter),
                               _ptr(iters), _ptr(trace), ws.ptr, _stream_ptr())
    _lib.check(rc, "rlvi_estep_deep_f32")


def evaluate_batch(logits, labels, out=None, ws=None):
    """Plain cross-entropy mean and top-1 percentage of one batch in one streaming pass over the
    logits (utils.evaluate, deep-learning/utils.py:48-62): out fp32[4] = {mean CE, top-1 %, sum
    CE, hits}."""
    L = _lib.load()
    _require_gpu(logits, labels)
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    labels = labels.to(torch.int64).contiguous()
    B, C = logits.shape
    if out is None:
        out = torch.empty(4, dtype=torch.float32, device=logits.device)
    ws = ws or workspace(logits.device, B, B)
    fn = L.rlvi_mstep_fwd_bwd_f32 if logits.dtype == torch.float32 else L.rlvi_mstep_fwd_bwd_bf16
    _lib.check(fn(_ptr(logits), logits.stride(0), _ptr(labels), None, None, None, B, B, C, 1.0 / B,
                  None, 0, _ptr(out), ws.ptr, _stream_ptr()), "rlvi_mstep_fwd_bwd (evaluation form)")
    return out


def mstep_reduce(scale=1.0, out=None, ws=None, device=None):
    """Collect (and clear) the records of accumulate-mode M-step calls -> out fp32[4]."""
    L = _lib.load()
    ws = ws or workspace(device or torch.cuda.current_device())
    if out is None:
        out = torch.empty(4, dtype=torch.float32, device=ws.buf.device)
    _lib.check(L.rlvi_mstep_reduce_f32(_ptr(out), float(scale), ws.ptr, _stream_ptr()),
               "rlvi_mstep_reduce_f32")
    return out


def _refuse_stale_peers(ws):
    if getattr(ws, "peers_stale", False):
        raise _lib.RlviError("a sharded call on this workspace timed out earlier: the ranks' round counters may "
                             "differ now -- run rlvi_amd.dist.setup_peers(ws) again on every rank first")


def estep_sharded(residuals, weights, n_all, tol=1e-3, maxiter=40, iters=None, ws=None, batches=0, out=None):
    """update_sample_weights (train_rlvi.py:14-38) with the samples sharded over the ranks: this rank's
    slice of residuals / weights in place, n_all samples over all ranks.  A collective over the ranks of
    the workspace's peer table (rlvi_amd.dist.setup_peers(ws) first).  Raises RlviError(RLVI_E_LIMIT)
    when the shape is outside the trajectory kernel.  batches > 0: `out` [4] also receives this rank's
    M-step scalars of the epoch (as epoch_end)."""
    L = _lib.load()
    _require_gpu(residuals, weights)
    for t in (residuals, weights):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
            raise ValueError("residuals / weights must be contiguous 1-D fp32 tensors")
    if ws is None:
        raise ValueError("the sharded E-step needs the workspace whose peer table was set up")
    _refuse_stale_peers(ws)
    rc = L.rlvi_estep_sharded_f32(_ptr(residuals), _ptr(weights), weights.shape[0], int(n_all), float(tol),
                                  int(maxiter), int(batches), _ptr(out) if batches > 0 else None, _ptr(iters),
                                  ws.ptr, _stream_ptr())
    _lib.check(rc, "rlvi_estep_sharded_f32")


def threshold_truncate_sharded(weights, n_all, threshold, alpha=0.05, want_mask=False, ws=None):
    """threshold_truncate on weights sharded over the ranks (the companion of estep_sharded): the
    criterion over ALL ranks' weights, max(threshold, criterion), truncation of this rank's weights.
    Returns (threshold, mask of this rank's weights or None, kept over all ranks) -- rank-identical."""
    L = _lib.load()
    _require_gpu(weights)
    if weights.dtype != torch.float32 or not weights.is_contiguous() or weights.dim() != 1:
        raise ValueError("weights must be a contiguous 1-D fp32 tensor")
    if ws is None:
        raise ValueError("the sharded threshold needs the workspace whose peer table was set up")
    _refuse_stale_peers(ws)
    thr = torch.as_tensor(threshold, dtype=torch.float32, device=weights.device).reshape(1).clone()
    mask = torch.empty(weights.shape[0], dtype=torch.uint8, device=weights.device) if want_mask else None
    kept = torch.zeros(1, dtype=torch.int64, device=weights.device)
    _lib.check(L.rlvi_threshold_truncate_sharded_f32(_ptr(weights), weights.shape[0], int(n_all), float(alpha),
                                                     _ptr(thr), _ptr(mask), _ptr(kept), ws.ptr, _stream_ptr()),
               "rlvi_threshold_truncate_sharded_f32")
    return thr.reshape(()), (mask.bool() if want_mask else None), kept.reshape(())


def epoch_end(residuals, weights, overfit=False, threshold=0, batches=0, tol=1e-3, maxiter=40,
              alpha=0.05, out=None, iters=None, ws=None):
    """train_rlvi.py:99-105 in one call: E-step over all samples, truncation when `overfit`, and
    the epoch's M-step scalars (out[1] = train_acc in percent when batches > 0).

    Returns (threshold, out): threshold is passed through unchanged while overfit is False, and
    a 0-dim device tensor after truncation ran; out is None when batches == 0."""
    L = _lib.load()
    _require_gpu(residuals, weights)
    for t in (residuals, weights):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
            raise ValueError("residuals / weights must be contiguous 1-D fp32 tensors")
    N = weights.shape[0]
    ws = ws or workspace(weights.device, N, 0)
    thr = None
    if overfit:
        thr = torch.as_tensor(threshold, dtype=torch.float32, device=weights.device).reshape(1).clone()
    if batches > 0 and out is None:
        out = torch.empty(4, dtype=torch.float32, device=weights.device)
    rc = L.rlvi_epoch_end_f32(_ptr(residuals), _ptr(weights), N, float(tol), int(maxiter),
                              1 if overfit else 0, float(alpha), _ptr(thr), int(batches),
                              _ptr(out) if batches > 0 else None, _ptr(iters), ws.ptr,
                              _stream_ptr())
    _lib.check(rc, "rlvi_epoch_end_f32")
    return (thr.reshape(()) if overfit else threshold), (out if batches > 0 else None)


def fn_threshold(weights, alpha=0.05, ws=None):
    """false_negative_criterion (train_rlvi.py:41-49) -> 0-dim fp32 device tensor."""
    L = _lib.load()
    _require_gpu(weights)
    w = weights if (weights.dtype == torch.float32 and weights.is_contiguous()) \
        else weights.float().contiguous()
    thr = torch.empty(1, dtype=torch.float32, device=w.device)
    ws = ws or workspace(w.device, w.shape[0], 0)
    _lib.check(L.rlvi_fn_threshold_f32(_ptr(w), w.shape[0], float(alpha), _ptr(thr), ws.ptr,
                                       _stream_ptr()), "rlvi_fn_threshold_f32")
    return thr.reshape(())


def threshold_truncate(weights, threshold, alpha=0.05, want_mask=False, ws=None):
    """train_rlvi.py:102-103: threshold = max(threshold, criterion); weights[weights<thr] = 0.

    Returns (threshold 0-dim tensor, mask bool tensor or None, kept int64 0-dim tensor)."""
    L = _lib.load()
    _require_gpu(weights)
    if weights.dtype != torch.float32 or not weights.is_contiguous():
        raise ValueError("weights must be a contiguous fp32 vector (it is truncated in place)")
    N = weights.shape[0]
    thr = torch.as_tensor(threshold, dtype=torch.float32, device=weights.device).reshape(1).clone()
    mask = torch.empty(N, dtype=torch.uint8, device=weights.device) if want_mask else None
    kept = torch.zeros(1, dtype=torch.int64, device=weights.device)
    ws = ws or workspace(weights.device, N, 0)
    _lib.check(L.rlvi_threshold_truncate_f32(_ptr(weights), N, float(alpha), _ptr(thr), _ptr(mask),
                                             _ptr(kept), ws.ptr, _stream_ptr()),
               "rlvi_threshold_truncate_f32")
    return thr.reshape(()), (mask.bool() if want_mask else None), kept.reshape(())


def fused_em(logits, labels, pi, tol=1e-3, maxiter=40, inv_scale=None, want_grad=True, ws=None,
             out=None, grad=None, rows=None, iters=None):
    """In-batch E+M (online order): NLL -> E-step on this batch -> weighted loss + grad.

    `pi` [B] fp32 is read (first error only) and overwritten with the new posteriors.
    Returns (out[4], grad, loss_rows (min-shifted NLL), iters int32 tensor)."""
    L = _lib.load()
    _require_gpu(logits, labels, pi)
    if logits.dtype != torch.float32:
        logits = logits.float()
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    labels = labels.to(torch.int64).contiguous()
    B, C = logits.shape
    if out is None:
        out = torch.empty(4, dtype=torch.float32, device=logits.device)
    if want_grad and grad is None:
        grad = torch.empty((B, C), dtype=torch.float32, device=logits.device)
    if not want_grad:
        grad = None
    if rows is None:
        rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    if iters is None:
        iters = torch.zeros(1, dtype=torch.int32, device=logits.device)
    ws = ws or workspace(logits.device, B, B)
    rc = L.rlvi_fused_em_f32(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(rows), _ptr(pi),
                             B, C, float(inv_scale if inv_scale is not None else 1.0 / B),
                             float(tol), int(maxiter), _ptr(grad),
                             grad.stride(0) if grad is not None else 0, _ptr(out), _ptr(iters),
                             ws.ptr, _stream_ptr())
    _lib.check(rc, "rlvi_fused_em_f32")
    return out, grad, rows, iters


def update_weights_f64(losses, tol=1e-3, maxiter=100, online=False):
    """standard-learning/rlvi.py:8-20 (online=False) or online-learning/main.py:45-58."""
    L = _lib.load()
    _require_gpu(losses)
    l = losses.to(torch.float64).contiguous()
    out = torch.empty_like(l)
    iters = torch.zeros(1, dtype=torch.int32, device=l.device)
    ws = workspace(l.device, l.shape[0], 0)
    fn = L.rlvi_update_weights_online_f64 if online else L.rlvi_update_weights_f64
    _lib.check(fn(_ptr(l), l.shape[0], float(tol), int(maxiter), _ptr(out), _ptr(iters), ws.ptr,
                  _stream_ptr()), "rlvi_update_weights_f64")
    return out, iters


def wls_solve(X, y, w, theta=None):
    """theta = argmin sum w_i (y_i - x_i.theta)^2 on the device (rlvi.py:70-71,:79-80): weighted Gram
    matrix on the fp64 MFMA units + Cholesky solve.  X [n, d <= 63] of full column rank."""
    L = _lib.load()
    _require_gpu(X, y, w)
    X = X.to(torch.float64).contiguous()
    n, d = X.shape
    if theta is None:
        theta = torch.empty(d, dtype=torch.float64, device=X.device)
    ws = workspace(X.device, n, 0)
    _lib.check(L.rlvi_wls_solve_f64(_ptr(X), _ptr(y.to(torch.float64).contiguous()),
                                    _ptr(w.to(torch.float64).contiguous()), n, d, _ptr(theta),
                                    ws.ptr, _stream_ptr()), "rlvi_wls_solve_f64")
    return theta


def linreg_losses(X, y, theta, w):
    """rlvi.py:72-74: losses = 0.5 (y - X theta)^2 / sigma2, sigma2 = w.r / sum(w)."""
    L = _lib.load()
    _require_gpu(X, y, theta, w)
    X = X.to(torch.float64).contiguous()
    n, d = X.shape
    losses = torch.empty(n, dtype=torch.float64, device=X.device)
    s2 = torch.empty(1, dtype=torch.float64, device=X.device)
    ws = workspace(X.device, n, 0)
    _lib.check(L.rlvi_linreg_losses_f64(_ptr(X), _ptr(y.to(torch.float64).contiguous()),
                                        _ptr(theta.to(torch.float64).contiguous()),
                                        _ptr(w.to(torch.float64).contiguous()), n, d,
                                        _ptr(losses), _ptr(s2), ws.ptr, _stream_ptr()),
               "rlvi_linreg_losses_f64")
    return losses, s2.reshape(())


def logistic_nll(X, w, b):
    """online-learning/main.py:295-296,:84-85: -log sigmoid(X w + b)."""
    L = _lib.load()
    _require_gpu(X, w)
    X = X.to(torch.float64).contiguous()
    n, d = X.shape
    losses = torch.empty(n, dtype=torch.float64, device=X.device)
    _lib.check(L.rlvi_logistic_nll_f64(_ptr(X), _ptr(w.to(torch.float64).contiguous()), float(b),
                                       n, d, _ptr(losses), _stream_ptr()),
               "rlvi_logistic_nll_f64")
    return losses


def linear_regression_check(n, d):
    """True if rlvi_linear_regression_f64 takes an [n, d] design in its one-launch form."""
    return _lib.load().rlvi_linear_regression_check(int(n), int(d)) == 0


def linear_regression(X, y, maxiter=100, tol=1e-3, estep_tol=1e-3, estep_maxiter=100, theta=None, weights=None,
                      info=None, ws=None):
    """linear_regression(X, y, maxiter, tol) of standard-learning/rlvi.py:68-89 as ONE launch on device tensors
    (no host synchronisation here).  Returns (theta [d], weights [n], info int32[4] = {outer iterations, inner
    iterations of the last E-step, inner iterations in all, fallback}) -- info[3] == 1 (rank-deficient design):
    theta / weights were not written, compose wls_solve / linreg_losses / update_weights_f64 instead."""
    L = _lib.load()
    _require_gpu(X, y)
    if X.dtype != torch.float64 or y.dtype != torch.float64 or not X.is_contiguous() or not y.is_contiguous():
        raise ValueError("X [n, d] and y [n] must be contiguous fp64 device tensors")
    n, d = X.shape
    if theta is None:
        theta = torch.empty(d, dtype=torch.float64, device=X.device)
    if weights is None:
        weights = torch.empty(n, dtype=torch.float64, device=X.device)
    if info is None:
        info = torch.zeros(4, dtype=torch.int32, device=X.device)
    ws = ws or workspace(X.device, n, 0)
    _lib.check(L.rlvi_linear_regression_f64(_ptr(X), _ptr(y), n, d, int(maxiter), float(tol), float(estep_tol),
                                            int(estep_maxiter), _ptr(theta), _ptr(weights), _ptr(info), ws.ptr,
                                            _stream_ptr()), "rlvi_linear_regression_f64")
    return theta, weights, info


def sample_weight_online(X, coef, intercept=0.0, first=False, tol=1e-3, maxiter=100, out=None, losses=None,
                         iters=None):
    """One mini-batch of online-learning/main.py:293-297 in ONE launch on device tensors: residuals =
    -log sigmoid(X coef + intercept) (log 2 for the first batch), sample_weight = update_weights_rlvi(residuals).
    Returns (sample_weight [n], losses or None, iters or None)."""
    L = _lib.load()
    _require_gpu(X, coef)
    n, d = X.shape
    if not first and (X.dtype != torch.float64 or not X.is_contiguous() or coef.dtype != torch.float64):
        raise ValueError("X [n, d] and coef [d] must be contiguous fp64 device tensors")
    if out is None:
        out = torch.empty(n, dtype=torch.float64, device=X.device)
    _lib.check(L.rlvi_sample_weight_online_f64(_ptr(X), _ptr(coef), float(intercept), 1 if first else 0, n, d,
                                               float(tol), int(maxiter), _ptr(losses), _ptr(out), _ptr(iters),
                                               _stream_ptr()), "rlvi_sample_weight_online_f64")
    return out, losses, iters


def stream_copy(dst, src):
    """Measurement aid: flat nontemporal 16-B/lane copy of src into dst (same byte count, 16-byte multiples)."""
    nbytes = src.numel() * src.element_size()
    if dst.numel() * dst.element_size() != nbytes:
        raise ValueError("stream_copy: sizes differ")
    _lib.check(_lib.load().rlvi_stream_copy(_ptr(dst), _ptr(src), nbytes, _stream_ptr()), "rlvi_stream_copy")


class _WeightedCE(torch.autograd.Function):
    """loss = mean_i(pi[idx_i] * CE(logits_i, y_i)) with the residual scatter as a side effect
    (train_rlvi.py:89-94); backward hands out the gradient the fused kernel already wrote."""

    @staticmethod
    def forward(ctx, logits, labels, idx, weights, residuals, inv_scale):
        out, grad = mstep_fwd_bwd(logits.detach(), labels, idx, weights, residuals, inv_scale)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        (grad,) = ctx.saved_tensors
        return grad * g_loss.to(grad.dtype), None, None, None, None, None


def select_smallest(losses, k, out=None):
    """mask[i] = 1.0 for the k smallest losses, else 0.0 -- np.argsort(loss)[:k] of the small-loss
    baselines (train_usdnl.py:18-24, train_coteaching.py:18-30) as a weight vector; equal losses in
    index order.  One launch, no host round trip."""
    L = _lib.load()
    _require_gpu(losses)
    if losses.dtype != torch.float32 or not losses.is_contiguous() or losses.dim() != 1:
        raise ValueError("losses must be a contiguous 1-D fp32 tensor")
    if out is None:
        out = torch.empty_like(losses)
    _lib.check(L.rlvi_select_smallest_f32(_ptr(losses), losses.shape[0], int(k), _ptr(out),
                                          _stream_ptr()), "rlvi_select_smallest_f32")
    return out


def topk_hits(logits, labels, ks, out=None):
    """Rows of the batch whose label is among the k largest logits, for every k of `ks` (at most 8): the counts
    behind accuracy(logit, target, topk) of deep-learning/utils.py:65-79.  Returns a device int32 tensor [len(ks)]
    (no host synchronisation here).  RuntimeError when a k exceeds the number of classes, as torch.topk raises."""
    import ctypes
    L = _lib.load()
    _require_gpu(logits, labels)
    if logits.dim() != 2:
        raise ValueError("logits must be [B, C]")
    B, C = logits.shape
    ks = [int(k) for k in ks]
    if not ks or len(ks) > 8:
        raise ValueError("between one and eight values of k per call")
    if max(ks) > C or min(ks) < 1:
        raise RuntimeError("selected index k out of range")
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    if labels.dtype != torch.int64 or not labels.is_contiguous():
        labels = labels.to(torch.int64).contiguous()
    if out is None:
        out = torch.empty(len(ks), dtype=torch.int32, device=logits.device)
    karr = (ctypes.c_int32 * len(ks))(*ks)
    fn = L.rlvi_topk_hits_f32 if logits.dtype == torch.float32 else L.rlvi_topk_hits_bf16
    _lib.check(fn(_ptr(logits), logits.stride(0), _ptr(labels), B, C, karr, len(ks), _ptr(out), _stream_ptr()),
               "rlvi_topk_hits")
    return out


def per_sample_ce(logits, labels, ws=None):
    """F.cross_entropy(logits, labels, reduction='none') by the streaming kernel (forward only)."""
    B = logits.shape[0]
    rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    ones = torch.ones(B, dtype=torch.float32, device=logits.device)
    mstep_fwd_bwd(logits, labels, None, ones, rows, want_grad=False, ws=ws)
    return rows


class _SelectedCE(torch.autograd.Function):
    """loss = inv_scale * sum_i mask_i * CE(logits_i, y_i) for a fixed 0/1 mask; backward hands out
    the gradient the fused kernel wrote (zero rows for unselected samples)."""

    @staticmethod
    def forward(ctx, logits, labels, mask, inv_scale):
        out, grad = mstep_fwd_bwd(logits.detach(), labels, None, mask, None, inv_scale)
        ctx.save_for_backward(grad)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g_loss):
        (grad,) = ctx.saved_tensors
        return grad * g_loss.to(grad.dtype), None, None, None


def selected_cross_entropy(logits, labels, mask, inv_scale):
    return _SelectedCE.apply(logits, labels, mask, inv_scale)


def weighted_cross_entropy(logits, labels, idx, weights, residuals, inv_scale=None):
    """Autograd entry: returns (loss 0-dim tensor, out[4])."""
    return _WeightedCE.apply(logits, labels, idx, weights, residuals, inv_scale)
