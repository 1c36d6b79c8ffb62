"""Batch sharding of the RLVI hot path over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference is single-device (no distributed code at all), so the contract is
"N ranks produce what one device produces on the concatenated batch":

  M-step   rows are independent: every rank streams its own rows with inv_scale = 1/B_global,
           so the SUM of the per-rank logit gradients / losses equals the single-device result.
           No data-path collective (the model-gradient all-reduce belongs to DDP).
  E-step   needs population-wide min / mean / max.  Every rank scatters its rows' NLL into ITS
           replica of residuals[N]; ONE all-gather makes the replicas identical; then every rank
           runs the whole fixed point redundantly (bit-identical pi everywhere, zero further
           collectives).  For N <~ 1e6 that beats 20-40 latency-bound scalar all-reduces
           (the "epsilon-prior all-reduce" of BASELINE.json, kept as `estep_allreduce_scalars`).
           SHARDED alternative (setup_peers + ops.estep_sharded, the bench's N > 1 path): every
           rank keeps only its own samples; the E-step kernel's reducer workgroups write their
           per-node totals straight into the other ranks' inboxes over xGMI (IPC-mapped uncached
           device memory) -- no gather of the residuals, no collective call, one extra hop of a few
           microseconds per round; every rank reaches the same fixed point bit for bit.
  threshold / mask: replicated on the identical pi.

Everything here is host logic on top of torch collectives and works on CPU tensors too.
"""
import torch
import torch.distributed as dist


_INPLACE_GATHER = True
_RAGGED = False


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size(group=None):
    return dist.get_world_size(group) if is_dist() else 1


def set_ragged(flag=True):
    """Per-rank batches of one step may differ in length (a hand-rolled sharded loader; torch's
    DistributedSampler pads to equal lengths and does not need this).  train_rlvi then sums the
    batch size over the ranks every step (one small all-reduce + host sync per batch) instead of
    checking once per epoch that the shards were equal."""
    global _RAGGED
    _RAGGED = bool(flag)


def ragged():
    return _RAGGED


_SHARERS = {}


def declare_device_sharing(group=None):
    """Collective (once per process group; later calls return the cached count): how many ranks of the
    group drive the SAME physical GPU as this one?  The kernels whose workgroups wait for each other size
    their grids from an occupancy query that sees one process only; S processes that each take what the
    query promises can together ask for more workgroups than the device holds, and every grid then waits
    for slots another one occupies until the spin bound ends them (RLVI_ST_TIMEOUT).  The ranks compare
    (host name, PCI bus id of the current device) and every rank that shares its GPU with S - 1 others tells
    the library to take 1/S of the proven capacity (RLVI_DEVICE_SHARERS).  One process per GPU -- the normal
    case -- gives 1 and changes nothing.  An explicit RLVI_DEVICE_SHARERS in the environment wins.
    Called by train_rlvi, setup_peers and bench.py; on a machine without a GPU it returns 1."""
    if not is_dist():
        return 1
    key = id(group) if group is not None else 0
    if key in _SHARERS:
        return _SHARERS[key]
    import ctypes
    import os
    import socket
    ident = None
    if torch.cuda.is_available():
        from . import _lib
        buf = ctypes.create_string_buffer(64)
        if _lib.load().rlvi_device_pci_bus_id(buf, 64) == 0:
            ident = (socket.gethostname(), buf.value.decode())
    idents = [None] * dist.get_world_size(group)
    dist.all_gather_object(idents, ident, group=group)
    n = sum(1 for x in idents if x is not None and x == ident) if ident is not None else 1
    if ident is not None and "RLVI_DEVICE_SHARERS" not in os.environ:
        from . import _lib
        _lib.check(_lib.load().rlvi_tune_set(b"RLVI_DEVICE_SHARERS", max(n, 1)), "rlvi_tune_set")
    _SHARERS[key] = max(n, 1)
    return _SHARERS[key]


_OWNER = None


def set_owner_sharding(owned, ws, peers, n_all=None, with_scalars=True, maxiter=40):
    """Opt in to the collective-free epoch end of train_rlvi: every rank owns a FIXED set of samples
    (`owned`: 1-D int64 device tensor of their indexes; every sample owned by exactly one rank; the rank's
    loader yields only those) and keeps residuals / weights of its own samples only.  The E-step and the
    threshold then run sharded (ops.estep_sharded / threshold_truncate_sharded on the compacted owned
    entries: the kernels exchange their totals through `peers`' inboxes), nothing is gathered, and after
    the call only the OWNED entries of residuals / weights are meaningful on a rank.  `ws`: the workspace
    `peers = setup_peers(ws)` was set up on.  None switches back to the replicated form.
    n_all: the length of the caller's residuals / weights vectors (= the number of samples over all ranks);
    when given, the ranks' shares must add up to it (the E-step divides by it).  Shares may differ in
    length: the sharded solve runs the same workgroup size on every rank."""
    global _OWNER
    if owned is not None and is_dist():
        # a collective: a rank that cannot launch (share too small / too large for the sharded kernels on
        # the co-residency this process is entitled to) would leave the others waiting for it -- so every
        # rank asks the library now and learns every rank's answer
        from . import _lib
        from ._lib import RlviError
        mine = int(owned.numel())
        total = [None] * dist.get_world_size()
        dist.all_gather_object(total, mine)
        n_sum = sum(total)
        if n_all is not None and n_sum != int(n_all):
            raise RlviError(f"owner sharding: the ranks' shares {total} add up to {n_sum}, not to the "
                            f"{int(n_all)} samples of the vectors (every sample needs exactly one owner)")
        # (both sharded launches of the epoch end, with the caller's maxiter: the E-step AND the threshold pick
        #  their geometry per rank from the co-residency the process is entitled to)
        ok = (mine > 1024 and
              _lib.load().rlvi_estep_sharded_check(mine, n_sum, int(maxiter), 1 if with_scalars else 0) == 0 and
              _lib.load().rlvi_threshold_sharded_check(mine, n_sum) == 0)
        oks = [None] * dist.get_world_size()
        dist.all_gather_object(oks, bool(ok))
        if not all(oks):
            raise RlviError("owner sharding: the sharded E-step / threshold cannot launch on every rank "
                            f"(shares {total}, launchable {oks}; 1025 .. 2097152 samples per rank, fewer at "
                            "the upper end when several ranks share one GPU)")
    _OWNER = None if owned is None else (owned, ws, peers)


def owner_sharding():
    return _OWNER


def _host_staged(t, group=None):
    """gloo has no device-side all-gather: collectives on CUDA tensors are staged through the host."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def check_equal_shards(sizes, group=None):
    """Once per epoch: every rank ran the same number of batches of the same lengths (what
    inv_scale = 1/B_local under DDP's gradient averaging assumes).  One small all-gather."""
    if not is_dist():
        return
    from ._lib import RlviError
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    n = torch.tensor([len(sizes)], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(x.item()) for x in ns]
    if len(set(ns)) != 1:
        raise RlviError(f"train_rlvi: ranks ran different numbers of batches this epoch ({ns}); "
                        "shard the loader evenly (torch DistributedSampler does)")
    mine = torch.tensor(list(sizes), dtype=torch.int64, device=dev)
    alls = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(alls, mine, group=group)
    if any(not torch.equal(a, alls[0]) for a in alls):
        raise RlviError("train_rlvi: per-rank batch lengths differ within a step; either shard evenly "
                        "(torch DistributedSampler pads) or call rlvi_amd.dist.set_ragged(True)")


def mean_scalars(t, group=None):
    """Mean over the ranks of a small tensor of per-rank means (train_acc, mean batch loss)."""
    if is_dist():
        dist.all_reduce(t, group=group)
        t /= dist.get_world_size(group)
    return t


def rank0_value(x, group=None):
    """Rank 0's value of a host scalar on every rank (one small broadcast).  For quantities that steer
    control flow -- the epoch driver's val_acc and the `overfit` flag derived from it: under DDP a model with
    BatchNorm evaluates slightly differently on every rank (the running statistics are only broadcast at
    the start of a forward), and a discrete decision that differs between ranks would let one rank
    truncate its weights alone, or enter a sharded kernel alone and wait for the others."""
    if not is_dist():
        return x
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return float(t.item())


def shard_range(n, rank, world):
    """Contiguous, balanced [start, stop) of n rows for `rank` (first n % world ranks get +1)."""
    q, r = divmod(int(n), int(world))
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def global_batch(b_local, group=None):
    """Sum of the per-rank batch sizes (one tiny all-reduce; skip it when shards are equal)."""
    if not is_dist():
        return int(b_local)
    t = torch.tensor([int(b_local)], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return int(t.item())


def exchange_residuals_owned(residuals, start, stop, group=None, force=False):
    """Every rank owns the contiguous slice [start, stop) of equal length: one
    all_gather_into_tensor straight into the replicated vector (N = world * (stop-start)).
    force: issue the collective even in a one-rank group (rehearsing graph capture)."""
    if not (is_dist() or (force and dist.is_initialized())):
        return
    world = dist.get_world_size(group)
    n = stop - start
    if residuals.shape[0] != n * world:
        raise ValueError("owned exchange needs equal contiguous slices covering the vector")
    if residuals.is_cuda and dist.get_backend(group) == "gloo":
        # debugging on a box without RCCL peers (e.g. several ranks sharing one GPU): stage on host
        host = torch.empty(residuals.shape, dtype=residuals.dtype)
        dist.all_gather_into_tensor(host, residuals[start:stop].cpu(), group=group)
        residuals.copy_(host)
        return
    # RCCL / NCCL gather in place when the send buffer is this rank's own slice of the receive
    # buffer; if a torch build refuses overlapping tensors, fall back to a copy of the slice (once)
    global _INPLACE_GATHER
    if _INPLACE_GATHER:
        try:
            dist.all_gather_into_tensor(residuals, residuals[start:stop], group=group)
            return
        except (RuntimeError, ValueError):
            _INPLACE_GATHER = False
    dist.all_gather_into_tensor(residuals, residuals[start:stop].clone(), group=group)


def exchange_residuals(residuals, idx_local, group=None):
    """General form (DistributedSampler-style ownership): all-gather (index, value) pairs of the
    rows this rank visited since the last exchange and scatter them into the local replica.
    Shards may differ in length (ragged last batch): they are padded to the longest."""
    if not is_dist():
        return
    world = dist.get_world_size(group)
    dev = residuals.device
    staged = _host_staged(residuals, group)
    cdev = torch.device("cpu") if staged else dev
    idx_local = idx_local.to(dev)
    n_local = torch.tensor([idx_local.numel()], dtype=torch.int64, device=cdev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad_idx = torch.zeros(m, dtype=torch.int64, device=dev)
    pad_val = torch.zeros(m, dtype=residuals.dtype, device=dev)
    pad_idx[:idx_local.numel()] = idx_local
    pad_val[:idx_local.numel()] = residuals[idx_local]
    all_idx = torch.empty(world * m, dtype=torch.int64, device=cdev)
    all_val = torch.empty(world * m, dtype=residuals.dtype, device=cdev)
    dist.all_gather_into_tensor(all_idx, pad_idx.to(cdev), group=group)
    dist.all_gather_into_tensor(all_val, pad_val.to(cdev), group=group)
    all_idx, all_val = all_idx.to(dev), all_val.to(dev)
    for r, s in enumerate(sizes):
        residuals[all_idx[r * m:r * m + s]] = all_val[r * m:r * m + s]


def reduce_scalars(t, group=None):
    """SUM-all-reduce of a small tensor of per-rank partial sums, e.g. {sum pi*l, hits}
    (once per epoch for logging; the training math never needs it per batch)."""
    if is_dist():
        dist.all_reduce(t, group=group)
    return t


def estep_allreduce_scalars(residuals_local, weights_local, n_global, tol=1e-3, maxiter=40,
                            group=None):
    """Measured alternative (BASELINE.json's "scalar epsilon-prior all-reduce"): the vectors stay
    SHARDED by rank; per iteration one packed 2-float SUM all-reduce {sum pi', sum (pi'-pi)^2},
    plus one MIN before and one MAX after.  Elementwise math is plain torch (this path exists to
    price the K serial collectives against the single all-gather; it is not the product path).
    Returns the iteration count."""
    r, w = residuals_local, weights_local
    mn = r.min().reshape(1)
    if is_dist():
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    r.sub_(mn)
    e = torch.exp(-r)
    ratio = torch.tensor(0.95 / (1 - 0.95), dtype=r.dtype, device=r.device)
    it = 0
    for _ in range(maxiter):
        new = ratio * e / (1 + ratio * e)
        s = torch.stack([new.double().sum(), ((new - w).double() ** 2).sum()])
        if is_dist():
            dist.all_reduce(s, group=group)
        w.copy_(new)
        it += 1
        err = s[1].sqrt().to(r.dtype)
        avg = (s[0].to(r.dtype) / n_global)
        if bool(err < tol):
            break
        ratio = avg / (1 - avg)
    mx = w.max().reshape(1)
    if is_dist():
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    w.div_(mx)
    return it


class Peers:
    """The cross-GPU exchange of the sharded E-step: one inbox per rank (uncached device memory), mapped
    into every other rank's process through an IPC handle.  Keep the object alive as long as the workspace
    is used for sharded E-steps; close() unmaps and frees."""

    def __init__(self, ws, group=None):
        import ctypes
        from . import _lib
        from . import ops
        L = _lib.load()
        self._L = L
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if multi else 1
        self.rank = dist.get_rank(group) if multi else 0
        self.own, self.mapped = None, []
        self._ws = ws
        # ranks that share a GPU (debugging on a one-GPU box) split its co-residency between them
        declare_device_sharing(group)
        err, raw = None, None
        try:
            if self.world > 8:
                raise _lib.RlviError("the sharded E-step exchanges over at most 8 ranks (one node)")
            own = ctypes.c_void_p()
            _lib.check(L.rlvi_peer_alloc(ctypes.byref(own)), "rlvi_peer_alloc")
            self.own = own
            handle = ctypes.create_string_buffer(64)
            _lib.check(L.rlvi_peer_export(own, handle), "rlvi_peer_export")
            raw = (handle.raw, torch.cuda.current_device())
        except Exception as e:  # noqa: BLE001  (agreed on below: every rank raises or none)
            err = repr(e)
        handles = [raw]
        if multi:
            handles = [None] * self.world
            dist.all_gather_object(handles, raw, group=group)
        if err is None and all(h is not None for h in handles):
            try:
                ptrs = (ctypes.c_void_p * self.world)()
                for r in range(self.world):
                    if r == self.rank:
                        ptrs[r] = self.own.value
                    else:
                        # (ordinals are comparable when every process sees all GPUs, as under torchrun;
                        #  with one masked GPU per process they are all 0 and the open decides)
                        if handles[r][1] != torch.cuda.current_device() and L.rlvi_peer_can_access(handles[r][1]) == 0:
                            raise _lib.RlviError(f"GPU {torch.cuda.current_device()} cannot map the memory of GPU {handles[r][1]}")
                        p = ctypes.c_void_p()
                        _lib.check(L.rlvi_peer_open(handles[r][0], ctypes.byref(p)), "rlvi_peer_open")
                        self.mapped.append(p)
                        ptrs[r] = p.value
                _lib.check(L.rlvi_workspace_set_peers(ws.ptr, self.rank, self.world, ptrs, ops._stream_ptr()),
                           "rlvi_workspace_set_peers")
                ws.peers_stale = False
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        elif err is None:
            err = "a peer could not export its inbox"
        errs = [err]
        if multi:
            # also the barrier: nobody pushes into an inbox that is not mapped everywhere yet
            errs = [None] * self.world
            dist.all_gather_object(errs, err, group=group)
        if any(e is not None for e in errs):
            self.close()
            raise _lib.RlviError("peer set-up failed: " + "; ".join(f"rank {r}: {e}" for r, e in enumerate(errs) if e))

    def close(self):
        ws = getattr(self, "_ws", None)
        if ws is not None and getattr(ws, "ptr", None) is not None and (self.mapped or self.own is not None):
            try:        # sharded calls on this workspace are refused from now on (no stale addresses)
                from . import ops
                self._L.rlvi_workspace_clear_peers(ws.ptr, ops._stream_ptr())
            except Exception:  # noqa: BLE001  (tearing down: nothing to report to)
                pass
        self._ws = None
        for p in self.mapped:
            self._L.rlvi_peer_close(p)
        self.mapped = []
        if self.own is not None:
            self._L.rlvi_peer_free(self.own)
            self.own = None


def setup_peers(ws, group=None):
    """Collective: allocate, exchange and map the inboxes and write the peer table of `ws`."""
    return Peers(ws, group)
