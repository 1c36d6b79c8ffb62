"""Host-side sanitizer pass over the CPU oracle (GPU AddressSanitizer is not available on this
pool, so the sanitizers run on the CPU build only): the C restatement is rebuilt with
-fsanitize=address,undefined and driven through its whole API on ragged / tiny / ties-heavy
inputs in a child process with libasan preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle import rlvi_oracle as O
O._LIB_PATH = %(lib)r
O._lib = None
from rlvi_amd import synth
rng = np.random.default_rng(0)
for (B, C) in ((1, 1), (3, 2), (17, 10), (64, 100), (33, 101)):
    d = synth.mstep_inputs(B, C, N=B + 5, seed=B)
    O.mstep(d["logits"], d["labels"], d["idx"], d["weights"], d["residuals"])
    O.nll_rows(d["logits"], d["labels"])
for N in (1, 2, 7, 64, 1000):
    for kind in ("equal", "exp", "heavy", "zeros10"):
        r = synth.residual_vector(kind, N, seed=N)
        w = np.ones(N, np.float32)
        O.update_sample_weights(r, w)
        thr = O.false_negative_criterion(w)
        O.truncate(w, thr)
        l = r.astype(np.float64)
        O.update_weights(l); O.update_weights_rlvi(l)
X, y = synth.linreg_data(40, 10, seed=1)
O.linreg_losses(X, y, np.ones(10), np.ones(40))
Xl, wl, b = synth.logistic_data(16, 5)
O.logistic_nll(Xl, wl, b)
print("sanitized-ok")
'''


@pytest.mark.timeout(300)
def test_oracle_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    lib = str(tmp_path / "librlvi_oracle_asan.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-fPIC", "-shared", "-std=c11", "-o", lib,
                           os.path.join(ROOT, "oracle", "rlvi_oracle.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1", PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=lib)], env=env,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "sanitized-ok" in p.stdout, p.stderr[-3000:]
