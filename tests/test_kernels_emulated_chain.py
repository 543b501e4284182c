"""conv_chain_kernel.h (2 or 4 chained 3x3 convolutions, the activated image handed over in LDS) on the CPU - MFMA, LDS-DMA and lane
exchanges emulated (tests/fakehip/README.md). A file of its own so that the CPU suite's workers share the emulated kernels' time; the
builder and the rewrite rules live in test_engine_emulated.py."""
import json
import os
import sys

from conftest import REPO
from test_engine_emulated import emu_full_lib, run_parallel  # noqa: F401  (emu_full_lib is a fixture)


CHAIN_CODE = r"""
import sys, json, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn
n_conv, X, Y, batch = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rng = np.random.default_rng(11)
cells = batch * X * Y
mask = np.ones((batch, Y, X), np.float32)
if batch > 1: mask[1, :, X - 3:] = 0; mask[1, Y - 2:, :] = 0   # a smaller board inside the buffer
m = mask.reshape(-1)
x = (rng.normal(size=(cells, 192)) * m[:, None]).astype(np.float32)
r = rng.normal(size=(cells, 192)).astype(np.float32)
w = (rng.normal(size=(n_conv, 192, 192, 3, 3)) * 0.03).astype(np.float32)
scale = rng.uniform(0.6, 1.4, (n_conv, 192)).astype(np.float32)
bias = rng.normal(0, 0.25, (n_conv, 192)).astype(np.float32)
out = {}
for dtype in sys.argv[6].split(","):
    res = {}
    for chained in (0, 2, 4):
        if chained > n_conv: continue
        R, Xo = nn.testEvaluateConvChain(batch, X, Y, dtype, x, r, w, scale, bias, 2, m, chained)
        res[chained] = (R, Xo)
    # torch restatement with the device's rounding points (16-bit tensors between layers, fp32 accumulation)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    q = lambda t: t.to(tdt).float()
    mish = lambda t: t * torch.tanh(torch.nn.functional.softplus(t))
    mt = torch.from_numpy(mask)[:, None]
    tx = q(torch.from_numpy(x.reshape(batch, Y, X, 192)).permute(0, 3, 1, 2))
    tr = q(torch.from_numpy(r.reshape(batch, Y, X, 192)).permute(0, 3, 1, 2))
    for k in range(0, n_conv, 2):
        sc = lambda i: torch.from_numpy(scale[i])[None, :, None, None]
        bi = lambda i: torch.from_numpy(bias[i])[None, :, None, None]
        t = q(mish(torch.nn.functional.conv2d(tx, q(torch.from_numpy(w[k])), padding=1) * sc(k) + bi(k)) * mt)
        v = torch.nn.functional.conv2d(t, q(torch.from_numpy(w[k + 1])), padding=1) + tr
        tr = q(v)
        tx = q(mish(v * sc(k + 1) + bi(k + 1)) * mt)
    wantR = tr.permute(0, 2, 3, 1).reshape(cells, 192).numpy()
    wantX = tx.permute(0, 2, 3, 1).reshape(cells, 192).numpy()
    R0, X0 = res[0]
    out[dtype] = {"same": {str(c): [bool(np.array_equal(res[c][0], R0)), bool(np.array_equal(res[c][1], X0))] for c in res if c},
                  "err": [float(np.abs(R0 - wantR).max()), float(np.abs(X0 - wantX).max())], "scale": [float(np.abs(wantR).max()), float(np.abs(wantX).max())],
                  "off_board_zero": bool((X0[m != 1.0] == 0).all()),
                  "digest": hashlib.sha1(R0.tobytes() + X0.tobytes()).hexdigest()}
print("RESULT " + json.dumps(out))
""" % (REPO, os.path.join(REPO, "tests"))


def test_convolution_chain_kernel_emulated(emu_full_lib):
    """conv_chain_kernel.h on the CPU (MFMA, LDS-DMA and lane exchanges emulated): one and two residual blocks on a 192-channel
    stream as chained launches of 2 and of 4 convolutions - the activated image handed over in LDS (chunks 0-2) and through the
    scratch tensor (chunks 3-5) - BIT FOR BIT against one launch of conv_kernel.h per convolution, and against a torch restatement
    with the device's rounding points; a 9x9 buffer with a smaller board in it (mask, halo zeros), a rectangular 13x7 buffer; with
    immediate LDS-DMA copies (a hand-over written into a slot that some wave still reads shows as a wrong answer) and with the
    latest completion the kernel's waits allow (a wait that does not cover a slab, an image chunk or the scratch stores leaves
    stale data)."""
    # (chain length, X, Y, boards, late completion, precisions); 19 x 19: the column order of boards at least 16 wide
    # (late completion 1: a copy lands at the wait that requires it; 2: at the barrier after that wait - see hip_runtime.h)
    cases = [("4", "9", "9", "2", "0", "bf16"), ("4", "9", "9", "2", "1", "fp16"), ("2", "13", "7", "1", "2", "bf16"),
             ("2", "19", "19", "1", "0", "fp16"), ("4", "19", "19", "1", "2", "bf16")]
    runs = run_parallel([([sys.executable, "-c", CHAIN_CODE, emu_full_lib, nc, X, Y, b, dt], dict(os.environ, KMX_EMU_LATE_DMA=late)) for nc, X, Y, b, late, dt in cases])
    for case, (rc, so, se) in zip(cases, runs):
        assert rc == 0 and "RESULT " in so, (case, (so + se)[-3000:])
        res = json.loads(so.split("RESULT ")[1])
        print(case, res)
        for dtype, r in res.items():
            assert all(all(v) for v in r["same"].values()) and r["off_board_zero"], (case, dtype, r)
            ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
            for e, sc in zip(r["err"], r["scale"]):
                assert e <= 6 * ulp * max(sc, 1.0), (case, dtype, r)
