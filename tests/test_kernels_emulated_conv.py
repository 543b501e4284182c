"""conv_kernel.h itself on the CPU - MFMA, LDS-DMA and lane exchanges emulated (tests/fakehip/README.md): layers against conv2d, a whole
small net with every launch through the real kernel against the oracle, the 8-wave shapes. A file of its own so that the CPU suite's workers
share the emulated kernels' time; the builder and the rewrite rules live in test_engine_emulated.py."""
import json
import os
import subprocess
import sys

from conftest import REPO
from test_engine_emulated import emu_full_lib, run_parallel  # noqa: F401  (emu_full_lib is a fixture)


def test_real_convolution_kernel_emulated(emu_full_lib):
    """conv_kernel.h itself on the CPU: 1x1 / 3x3 / 5x5, 4-wave and 8-wave shapes, channel counts that are not multiples of
    the tile, a masked small board — against torch.nn.functional.conv2d on the 16-bit-rounded operands; then a whole small
    net (every launch through the real kernel) against the oracle."""
    code = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn, modelgen
from oracle import oracle
from conftest import make_rows
rng = np.random.default_rng(0)
out = {}
for (ks, cin, cout, X, Y, n) in ((3, 32, 32, 9, 9, 1), (1, 64, 96, 19, 19, 1), (3, 40, 200, 19, 19, 1), (5, 22, 64, 13, 9, 1), (3, 64, 128, 19, 19, 2)):
    w = (rng.normal(size=(cout, cin, ks, ks)) * 0.1).astype(np.float32)
    x = rng.normal(size=(n, Y * X, cin)).astype(np.float32)
    got = np.asarray(nn.testEvaluateConv(w, n, X, Y, True, x))
    xt = torch.from_numpy(x.reshape(n, Y, X, cin).transpose(0, 3, 1, 2)).to(torch.bfloat16).float()
    wt = torch.from_numpy(w).to(torch.bfloat16).float()
    want = torch.nn.functional.conv2d(xt, wt, padding=ks // 2).numpy().transpose(0, 2, 3, 1).reshape(n, Y * X, cout)
    out["conv%%d_%%d_%%d" %% (ks, cin, cout)] = [float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(want).max())]
nn.globalInitialize()
ctx = nn.createComputeContext([0], 19, 19, precision="bf16")
p = "/tmp/kmx_emufull_b2c32nbt.bin"
modelgen.write_model(p, "b2c32nbt", seed=4)
sp, gl = make_rows(rng, 2, 19, [(19, 19), (9, 13)])
sym = np.array([3, 6], np.int32); opt = np.array([0.0, 1.0], np.float32)
h = nn.createComputeHandle(ctx, nn.loadModelFile(p), 2)
got = nn.getOutput(h, sp, gl, sym, opt)
want = oracle.getOutput(oracle.loadModelFile(p), 19, 19, sp, gl, sym, opt)
mask = sp[:, :, 0] > 0; full = np.concatenate([mask, np.ones((2, 1), bool)], axis=1)
out["net"] = {"policy": [float(np.abs(got["policy"] - want["policy"])[full].max()), float(np.abs(want["policy"][full]).max())],
              "value": [float(np.abs(got["value"] - want["value"]).max()), float(np.abs(want["value"]).max())],
              "ownership": [float(np.abs(got["ownership"] - want["ownership"])[mask].max()), float(np.abs(want["ownership"][mask]).max())]}
print("RESULT " + json.dumps(out))
""" % (REPO, os.path.join(REPO, "tests"))
    p = subprocess.run([sys.executable, "-c", code, emu_full_lib], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    res = json.loads(p.stdout.split("RESULT ")[1])
    for k, v in res.items():
        if k.startswith("conv"):
            assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, (k, v)  # one bf16 rounding of the output
    for k, (err, scale) in res["net"].items():
        assert err <= 0.08 + 0.03 * scale, (k, err, scale)


def test_eight_wave_shapes_emulated(emu_full_lib):
    """The product's 8-wave 3x3 shapes forced at a small batch with KMX_CONV_TUNE=min_wgs8=1: same answers as conv2d. (The even-tap-barrier
    variant that this test also ran in rounds 2-3 spilled registers on the hardware and is deleted; DESIGN.md 4.8 keeps the record.)"""
    code = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn
rng = np.random.default_rng(1)
out = {}
for (cin, cout, X, Y, n) in ((64, 192, 19, 19, 1), (96, 128, 13, 9, 1)):
    w = (rng.normal(size=(cout, cin, 3, 3)) * 0.1).astype(np.float32)
    x = rng.normal(size=(n, Y * X, cin)).astype(np.float32)
    got = np.asarray(nn.testEvaluateConv(w, n, X, Y, True, x))
    xt = torch.from_numpy(x.reshape(n, Y, X, cin).transpose(0, 3, 1, 2)).to(torch.bfloat16).float()
    wt = torch.from_numpy(w).to(torch.bfloat16).float()
    want = torch.nn.functional.conv2d(xt, wt, padding=1).numpy().transpose(0, 2, 3, 1).reshape(n, Y * X, cout)
    out["%%d_%%d" %% (cin, cout)] = [float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(want).max())]
print("RESULT " + json.dumps(out))
""" % (REPO,)
    (rc, so, se), = run_parallel([([sys.executable, "-c", code, emu_full_lib], dict(os.environ, KMX_CONV_TUNE="min_wgs8=1"))])
    assert rc == 0, (so + se)[-3000:]
    for k, v in json.loads(so.split("RESULT ")[1]).items():
        assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, (k, v)
