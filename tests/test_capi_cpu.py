"""CPU-side checks of the C-ABI library and the host logic (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rlvi_amd import _build, _lib
    _build.build()
    return _lib.load()


def header_functions():
    src = open(os.path.join(ROOT, "include", "rlvi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rlvi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from rlvi_amd import _lib
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rlvi_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names      # the ctypes table covers the whole header


def test_abi_version_and_error_strings(lib):
    assert lib.rlvi_abi_version() == 3
    assert lib.rlvi_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert lib.rlvi_error_string(code) not in (b"ok", b"unknown rlvi error")


def test_workspace_bytes_monotone(lib):
    a = lib.rlvi_workspace_bytes(0, 0)
    b = lib.rlvi_workspace_bytes(65536, 65536)
    c = lib.rlvi_workspace_bytes(1 << 20, 65536)
    assert 0 < a <= b <= c and a % 256 == 0 and c % 256 == 0


def test_argument_errors_without_a_gpu(lib):
    """Argument validation happens on the host before any launch."""
    assert lib.rlvi_estep_deep_f32(None, None, 10, 1e-3, 40, None, None, None, None) == -1
    assert lib.rlvi_fn_threshold_f32(None, 10, 0.05, None, None, None) == -1
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 255) & ~255
    assert lib.rlvi_estep_deep_f32(p, p, 0, 1e-3, 40, None, None, p, None) == -2
    assert lib.rlvi_estep_deep_f32(p, p, -5, 1e-3, 40, None, None, p, None) == -2
    assert lib.rlvi_estep_deep_f32(p, p + 2, 8, 1e-3, 40, None, None, p, None) == -3
    assert lib.rlvi_mstep_fwd_bwd_f32(p, 4, p, p, p, p, 8, 8, 10, 0.1, None, 0, p, p, None) == -2  # ld < C
    assert lib.rlvi_update_weights_f64(p, 0, 1e-3, 10, p, None, p, None) == -2


def test_knobs_can_be_taken_back_and_listed(lib):
    """rlvi_tune_set is process-wide: rlvi_tune_unset takes a value back (the environment / default applies again),
    rlvi_tune_overrides lists what is set -- what the GPU suite's autouse fixture asserts to be empty after every
    test.  Host-side only: no device needed."""
    from rlvi_amd import _lib
    for n in _lib.tune_overrides():
        lib.rlvi_tune_unset(n.encode())
    assert _lib.tune_overrides() == [] and lib.rlvi_tune_unset(b"RLVI_THR_LIST") == 0
    assert lib.rlvi_tune_set(b"RLVI_THR_LIST", 7) == 0 and lib.rlvi_tune_set(b"RLVI_FUSED_EM", 0) == 0
    assert sorted(_lib.tune_overrides()) == ["RLVI_FUSED_EM", "RLVI_THR_LIST"]
    assert lib.rlvi_tune_set(b"RLVI_THR_LIST", 9) == 0 and len(_lib.tune_overrides()) == 2     # same knob, new value
    assert lib.rlvi_tune_unset(b"RLVI_THR_LIST") == 1 and lib.rlvi_tune_unset(b"RLVI_THR_LIST") == 0
    assert _lib.tune_overrides() == ["RLVI_FUSED_EM"]
    assert lib.rlvi_tune_unset(b"RLVI_FUSED_EM") == 1 and _lib.tune_overrides() == []
    assert lib.rlvi_tune_unset(None) == -1


def test_workspace_regions_options_and_shape_queries(lib):
    """Host-side queries of ABI 3: the named regions of the workspace layout lie inside the smallest workspace, in
    order and apart; a per-workspace option is refused for an unknown name; the one-launch estimator says which
    shapes it takes."""
    n = ctypes.c_size_t(0)
    small = lib.rlvi_workspace_bytes(0, 0)
    offs = {}
    for name in (b"warm", b"records", b"records_out", b"scratch"):
        offs[name] = lib.rlvi_workspace_region(name, ctypes.byref(n))
        assert 0 < offs[name] < small, name
        if name != b"scratch":
            assert n.value > 0
    assert offs[b"warm"] < offs[b"records"] < offs[b"records_out"] < offs[b"scratch"]
    assert offs[b"records_out"] - offs[b"records"] >= 1024 * 4 * 8          # 1024 workgroups x 4 doubles each
    assert lib.rlvi_workspace_region(b"nope", None) == ctypes.c_size_t(-1).value
    buf = (ctypes.c_char * 512)()
    p = (ctypes.addressof(buf) + 255) & ~255
    assert lib.rlvi_workspace_set_option(p, b"logits_from_hbm", 1) == 0
    assert lib.rlvi_workspace_set_option(p, b"cold_start", 1) == 0
    assert lib.rlvi_workspace_set_option(p, b"no_such_option", 1) == -2
    assert lib.rlvi_workspace_set_option(None, b"cold_start", 1) == -1
    assert lib.rlvi_workspace_last_mstep_form(p) == 0
    assert lib.rlvi_linear_regression_check(1000, 20) == 0 and lib.rlvi_linear_regression_check(40, 10) == 0
    assert lib.rlvi_linear_regression_check(4096, 31) == 0
    assert lib.rlvi_linear_regression_check(4097, 5) == -5 and lib.rlvi_linear_regression_check(100, 32) == -5
    assert lib.rlvi_linear_regression_check(0, 5) == -2
    assert lib.rlvi_stream_copy(None, None, 16, None) == -1
    k15 = (ctypes.c_int32 * 2)(1, 5)
    assert lib.rlvi_topk_hits_f32(None, 10, p, 4, 10, k15, 2, p, None) == -1
    assert lib.rlvi_topk_hits_f32(p, 10, p, 4, 10, k15, 0, p, None) == -2
    assert lib.rlvi_topk_hits_f32(p, 10, p, 4, 10, k15, 9, p, None) == -2
    assert lib.rlvi_topk_hits_f32(p, 4, p, 4, 4, k15, 2, p, None) == -2            # k = 5 beyond C = 4: torch.topk raises
    assert lib.rlvi_topk_hits_bf16(p, 10, p + 4, 4, 10, k15, 2, p, None) == -3     # labels not 8-byte aligned
    assert lib.rlvi_stream_copy(p, p + 8, 16, None) == -3 and lib.rlvi_stream_copy(p, p + 16, 24, None) == -3


def test_ops_refuse_cpu_tensors():
    import torch
    from rlvi_amd import _lib, ops
    z = torch.zeros(4, 10)
    with pytest.raises(_lib.RlviError, match="no CPU fallback"):
        ops.mstep_fwd_bwd(z, torch.zeros(4, dtype=torch.int64), torch.arange(4), torch.ones(4),
                          torch.zeros(4))
    with pytest.raises(_lib.RlviError):
        ops.estep_deep(torch.zeros(4), torch.ones(4))
    with pytest.raises(_lib.RlviError):
        ops.fn_threshold(torch.ones(4))
    with pytest.raises(_lib.RlviError, match="no CPU fallback"):
        ops.MStepLoop(torch.ones(4), torch.zeros(4))          # the training loop's launcher: same rule


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from rlvi_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RlviError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under rlvi_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rlvi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "rlvi_oracle" not in txt, f


def test_plugin_interface_matches_reference_names():
    import inspect
    from rlvi_amd import methods
    from rlvi_amd.methods import train_rlvi as mod_fn
    import rlvi_amd.methods.train_rlvi  # noqa: F401
    import sys
    m = sys.modules["rlvi_amd.methods.train_rlvi"]
    assert m.__all__ == ['train_rlvi']
    assert list(inspect.signature(m.train_rlvi).parameters) == [
        "train_loader", "model", "optimizer", "residuals", "weights", "overfit", "threshold"]
    sig = inspect.signature(m.update_sample_weights)
    assert list(sig.parameters) == ["residuals", "weights", "tol", "maxiter"]
    assert sig.parameters["tol"].default == 1e-3 and sig.parameters["maxiter"].default == 40
    sig = inspect.signature(m.false_negative_criterion)
    assert list(sig.parameters) == ["weights", "alpha"] and sig.parameters["alpha"].default == 0.05
    assert methods.train_rlvi is mod_fn


def test_small_loss_baseline_interfaces_match_reference_names():
    """train_usdnl.py:16,30 and train_coteaching.py:17,39: same names, positional order, __all__."""
    import inspect
    import sys
    from rlvi_amd import methods
    import rlvi_amd.methods.train_coteaching  # noqa: F401
    import rlvi_amd.methods.train_usdnl  # noqa: F401
    u = sys.modules["rlvi_amd.methods.train_usdnl"]
    c = sys.modules["rlvi_amd.methods.train_coteaching"]
    assert u.__all__ == ['train_usdnl'] and c.__all__ == ['train_coteaching']
    assert list(inspect.signature(u.loss_fn).parameters) == ["logits", "labels", "forget_rate"]
    assert list(inspect.signature(u.train_usdnl).parameters) == [
        "train_loader", "epoch", "model", "optimizer", "rate_schedule"]
    assert list(inspect.signature(c.loss_coteaching).parameters) == ["y_1", "y_2", "t", "forget_rate", "ind"]
    assert list(inspect.signature(c.train_coteaching).parameters) == [
        "train_loader", "epoch", "model1", "optimizer1", "model2", "optimizer2", "rate_schedule"]
    assert methods.train_usdnl is u.train_usdnl and methods.train_coteaching is c.train_coteaching


def _build_capi_example(tmp_path):
    import shutil
    import subprocess
    from rlvi_amd import _build
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "capi_smoke")
    libdir = os.path.dirname(_build.LIB)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-I" + os.path.join(_build.ROOT, "include"),
                           os.path.join(_build.ROOT, "examples", "capi_smoke.cpp"), "-L" + libdir,
                           "-lrlvi_gfx950", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_cpp_host_program_links_against_the_c_abi(lib, tmp_path):
    """examples/capi_smoke.cpp (no Python, no torch) compiles and links against the header + .so."""
    assert os.path.exists(_build_capi_example(tmp_path))


def test_driver_models_and_schedule():
    """ResNet18 / LeNet shapes and the LR factors of utils.get_lr_factor (CPU-only checks)."""
    import torch
    from rlvi_amd import driver
    assert driver.get_lr_factor(0) == 1.0 and driver.get_lr_factor(20) == 1.0
    assert driver.get_lr_factor(30) == pytest.approx(1.0 - 0.99 * 0.5)
    assert driver.get_lr_factor(40) == pytest.approx(0.01) and driver.get_lr_factor(99) == 0.01
    net = driver.ResNet18(3, 10)
    assert net(torch.zeros(2, 3, 32, 32)).shape == (2, 10)
    assert sum(p.numel() for p in net.parameters()) == 11173962 - 0   # CIFAR ResNet18, 10 classes
    assert driver.LeNet(1, 10)(torch.zeros(2, 1, 28, 28)).shape == (2, 10)


def test_bench_plans_its_ranks_and_says_what_it_could_not_measure():
    """`python bench.py --gpus N` becomes N ranks itself; with fewer devices than asked for it runs on the
    visible ones and says so (never an M-GPU number in N's clothing).  Here (no GPU): the planning rule, and
    the end-to-end record of a box with no device at all -- one JSON line, n_gpus 0, exit code 3."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.plan_ranks(8, 8, False) == (8, None)
    assert bench.plan_ranks(2, 8, False) == (2, None)
    assert bench.plan_ranks(8, 1, False) == (1, "8 requested, 1 visible")
    assert bench.plan_ranks(4, 2, False) == (2, "4 requested, 2 visible")
    assert bench.plan_ranks(3, 1, True) == (3, None)            # --same-device: ranks share cuda:0
    assert bench.plan_ranks(2, 0, False) == (0, "2 requested, 0 visible")
    import torch
    if torch.cuda.device_count() == 0:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                            "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert p.returncode == 3, p.stderr[-1000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 0 and rec["value"] is None and rec["not_measured"] == "2 requested, 0 visible"


def test_package_import_selects_the_dmabuf_ipc_mode(monkeypatch):
    """Multi-process GPU work (RCCL, the peers' inboxes) needs HSA_ENABLE_IPC_MODE_LEGACY=0 before the runtime
    starts: importing the package defaults it and never overrides an explicit setting."""
    import importlib
    import rlvi_amd
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    importlib.reload(rlvi_amd)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    importlib.reload(rlvi_amd)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
