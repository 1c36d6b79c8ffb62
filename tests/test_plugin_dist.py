"""Host logic of the plug-in (rlvi_amd/methods/train_rlvi.py) that does not need a GPU: the sticky
device status must raise, and under a torch.distributed group (two gloo ranks on 127.0.0.1, model in
DistributedDataParallel) train_rlvi must reproduce the reference's four golden epochs (G4) with the
batch split over the ranks.  The HIP kernels are replaced by the CPU oracle (tests/ops_standin.py);
the same scenario with the real kernels, two ranks sharing cuda:0, is in test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "g4_epoch.npz")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_g4_epochs(train_rlvi, device, rank, world, wrap=None):
    """Four epochs of G4 with every batch split into `world` equal contiguous shards; returns the
    per-epoch state (numpy) of this rank."""
    g = np.load(GOLDEN)
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    N, B = int(g["N"]), int(g["B"])
    model = torch.nn.Linear(X.shape[1], 10)
    with torch.no_grad():
        model.weight.copy_(torch.from_numpy(g["W0"]))
        model.bias.copy_(torch.from_numpy(g["b0"]))
    model.to(device)
    net = wrap(model) if wrap is not None else model
    opt = torch.optim.SGD(net.parameters(), lr=float(g["lr"]), momentum=float(g["momentum"]))
    residuals = torch.zeros(N, device=device)
    weights = torch.ones(N, device=device)
    threshold = 0
    states = []
    for ep in range(4):
        perm = g["orders"][ep]
        loader = []
        for s in range(0, N, B):
            rows = perm[s:s + B]
            n = len(rows) // world
            rows = rows[rank * n:(rank + 1) * n]
            loader.append((X[rows], y[rows], torch.from_numpy(rows.astype(np.int64))))
        net.train()
        acc, threshold = train_rlvi(loader, net, opt, residuals, weights, bool(g[f"ep{ep}/overfit"]), threshold)
        states.append(dict(W=model.weight.detach().cpu().numpy().copy(),
                           residuals=residuals.cpu().numpy().copy(), weights=weights.cpu().numpy().copy(),
                           threshold=float(threshold), acc=float(acc)))
    return states


G4_DRIFT_MULT = 16.0      # accepted distance from G4 in units of the reference's own fp32-vs-fp64 drift


def check_against_g4(states):
    """G4 holds, next to the reference's fp32 outputs, how far they drift from the same epochs run in
    fp64 (drift/*).  Accept G4_DRIFT_MULT of that drift plus one fp32 ulp of the largest value."""
    g = np.load(GOLDEN)
    for ep, st in enumerate(states):
        for name in ("W", "residuals", "weights"):
            ref = g[f"ep{ep}/{name}"]
            tol = G4_DRIFT_MULT * float(g[f"drift/ep{ep}/{name}"]) + 1.2e-7 * float(np.abs(ref).max())
            err = float(np.abs(st[name] - ref).max())
            assert err <= tol, (ep, name, err, tol)
        thr_tol = G4_DRIFT_MULT * max(float(g[f"drift/ep{ep}/threshold"]), float(g[f"drift/ep{ep}/weights"])) + 1.2e-7
        assert abs(st["threshold"] - float(g[f"ep{ep}/threshold"])) <= thr_tol
        assert st["acc"] == pytest.approx(float(g[f"ep{ep}/train_acc"]), abs=1e-3)


def test_nonzero_device_status_raises_through_train_rlvi(monkeypatch, oracle):
    """RLVI_ST_TIMEOUT / NOCONV / RANGE left by a kernel must not pass silently (the weights would be
    stale or NaN): train_rlvi reads the status word at its one host sync and raises."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ops_standin
    import rlvi_amd.methods.train_rlvi  # noqa: F401
    from rlvi_amd import _lib
    mod = sys.modules["rlvi_amd.methods.train_rlvi"]
    for status, word in ((2, "timed out"), (4, "fixed point"), (1, "out of range")):
        fake = ops_standin.StandIn(status=status)
        monkeypatch.setattr(mod, "ops", fake)
        with pytest.raises(_lib.RlviError, match=word):
            run_g4_epochs(mod.train_rlvi, torch.device("cpu"), 0, 1)
        assert fake.ws.status() == 0            # cleared: the caller can recover and go on
    monkeypatch.setattr(mod, "ops", ops_standin.StandIn(status=0))
    check_against_g4(run_g4_epochs(mod.train_rlvi, torch.device("cpu"), 0, 1))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import ops_standin
        import rlvi_amd.methods.train_rlvi  # noqa: F401
        from oracle import rlvi_oracle as O
        O.set_threads(1)
        mod = sys.modules["rlvi_amd.methods.train_rlvi"]
        mod.ops = ops_standin.StandIn()
        from torch.nn.parallel import DistributedDataParallel as DDP
        states = run_g4_epochs(mod.train_rlvi, torch.device("cpu"), rank, world, wrap=lambda m: DDP(m))
        check_against_g4(states)
        # rank-identical pi, threshold and train_acc
        for st in states:
            flat = torch.from_numpy(np.concatenate([st["weights"], [st["threshold"], st["acc"]]]).astype(np.float64))
            both = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            assert all(torch.equal(b, both[0]) for b in both)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(e) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_train_rlvi_two_ranks_ddp_reproduces_g4(oracle):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in results), results


def _worker_unequal(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rlvi_amd import _lib
        from rlvi_amd import dist as rdist
        try:
            rdist.check_equal_shards([64, 64, 32 + rank])
            q.put((rank, "FAIL no error"))
        except _lib.RlviError:
            rdist.check_equal_shards([64, 64, 32])
            q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL " + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_unequal_shards_are_refused():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unequal, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in results), results


def _worker_peers_without_gpu(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rlvi_amd import _lib
        from rlvi_amd import dist as rdist

        class NoWs:
            ptr = None
        try:
            rdist.setup_peers(NoWs())
            q.put((rank, "FAIL no error"))
        except _lib.RlviError as e:
            # both ranks learn about both failures and leave the set-up together: the group still works
            assert "rank 0" in str(e) and "rank 1" in str(e), str(e)
            dist.barrier()
            q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL " + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_peer_setup_fails_loudly_and_on_every_rank_without_a_gpu():
    """The sharded E-step's inboxes are device memory: on a box without a GPU the set-up must raise the
    same RlviError on every rank (the ranks agree on the failure before anyone leaves), never hang and
    never fall back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_peers_without_gpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in results), results


def _worker_driver_overfit(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rlvi_amd import driver
        # what BatchNorm's per-rank running statistics can do to a replica's evaluation: the ranks see
        # slightly different accuracies -- here so different that rank 1 alone would never set `overfit`
        seq = {0: [50.0, 60.0, 70.0, 30.0, 30.0, 30.0, 30.0], 1: [50.0, 60.0, 70.0, 71.0, 72.0, 73.0, 74.0]}[rank]
        calls = {"n": 0}
        flags = []

        def evaluate_fn(loader, model, device):
            # (called for test, then per epoch val + test: the val calls are the odd ones)
            calls["n"] += 1
            k = calls["n"]
            return seq[min((k - 2) // 2, len(seq) - 1)] if k >= 2 and k % 2 == 0 else 10.0 + rank

        def train_fn(loader, net, opt, residuals, weights, overfit, threshold):
            flags.append(bool(overfit))
            return 0.0, threshold
        logs = driver.run(n_train=64, n_val=16, n_test=16, batch_size=16, n_epoch=7, device="cpu",
                          train_fn=train_fn, evaluate_fn=evaluate_fn, schedule=False)
        mine = torch.tensor([float(f) for f in flags] + [r["val_acc"] for r in logs] + [r["test_acc"] for r in logs],
                            dtype=torch.float64)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1]), (both[0], both[1])      # same decisions, same log on every rank
        assert flags == [False, False, False, False, True, True], flags   # rank 0's numbers: overfit after epoch 4
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(e) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_epoch_driver_takes_rank_identical_overfit_decisions():
    """The driver's `overfit` switch (main.py:283-288) is a discrete decision on val_acc; under DDP every
    rank evaluates its own replica, and replicas with BatchNorm differ.  Rank 0's value steers every rank:
    with evaluation functions that disagree between the ranks, both ranks pass the same `overfit` flags to
    the training function and log the same accuracies."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_driver_overfit, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in results), results
