"""The seam kernels (pointwise_kernel.h, pointwise2_kernel.h, pointwise3_kernel.h) themselves on the CPU - MFMA, LDS-DMA and lane
exchanges emulated (tests/fakehip/README.md), with immediate and with the latest legal completion of their asynchronous copies. A file of
its own so that the CPU suite's workers share the emulated kernels' time; the builder, the rewrite rules and PW2_CODE live in
test_engine_emulated.py."""
import json
import os
import sys

from conftest import REPO
from test_engine_emulated import PW2_CODE, emu_full_lib, run_parallel  # noqa: F401  (emu_full_lib is a fixture)


def test_pointwise_seam_kernel_emulated(emu_full_lib):
    """pointwise_kernel.h itself on the CPU (MFMA and LDS-DMA emulated): one full 128-cell tile plus a tail tile (2 boards of
    9x9 = 162 cells), masked cells, against the numpy restatement and bit for bit against the two emulated convolution launches."""
    code = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn
import pointwise_ref as ref
rng = np.random.default_rng(3)
batch, L = 2, 9
mask = np.ones((batch, L, L), np.float32); mask[1, :, 6:] = 0
x, resid, w1, s1, b1, w2, s2, b2, m = ref.make_case(rng, batch * L * L, 192, 384, 192, mask.reshape(-1))
out = {}
for dtype in ("bf16", "fp16"):
    fused = nn.testEvaluatePointwisePair(batch, L, L, dtype, x, resid, w1, s1, b1, 2, w2, s2, b2, 1, m, True)
    plain = nn.testEvaluatePointwisePair(batch, L, L, dtype, x, resid, w1, s1, b1, 2, w2, s2, b2, 1, m, False)
    want = ref.seam(x, resid, w1, s1, b1, 2, w2, s2, b2, 1, m, dtype)
    out[dtype] = {"same": [bool(np.array_equal(f, p)) for f, p in zip(fused, plain)],
                  "err": [float(np.abs(f - w).max()) for f, w in zip(fused, want)],
                  "scale": [float(np.abs(w).max()) for w in want],
                  "off_board_zero": bool((fused[2][m != 1.0] == 0).all())}
print("RESULT " + json.dumps(out))
""" % (REPO, os.path.join(REPO, "tests"))
    # the product shape (8 waves x 128 cells) and the two-per-CU experiment (4 waves x 64 cells); each with immediate LDS-DMA
    # copies and with the latest completion its s_waitcnt counts allow (KMX_EMU_LATE_DMA=1, tests/fakehip/emul/hip/hip_runtime.h)
    variants = (("8", "0"), ("4", "0"), ("8", "2"), ("4", "1"))
    runs = run_parallel([([sys.executable, "-c", code, emu_full_lib], dict(os.environ, KMX_PW_WAVES=w, KMX_PW_V2="0", KMX_EMU_LATE_DMA=late))
                         for w, late in variants])
    for waves, (rc, so, se) in zip(variants, runs):
        assert rc == 0 and "RESULT " in so, (so + se)[-3000:]
        res = json.loads(so.split("RESULT ")[1])
        print(waves, res)
        for dtype, r in res.items():
            assert all(r["same"]) and r["off_board_zero"], (waves, dtype, r)
            ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
            for e, s, k in zip(r["err"], r["scale"], (1, 4, 4)):
                assert e <= 2 * ulp * max(s, 1.0) * k, (waves, dtype, r)


def test_persistent_seam_kernel_emulated(emu_full_lib):
    """pointwise2_kernel.h on the CPU: a work-group that walks three tiles (two full, one tail of 82 cells; KMX_PW_GRID=1) and two
    work-groups that share them (KMX_PW_GRID=2) - the ring-slot reuse, the part order and the prefetch of the next tile's X are
    all exercised with IMMEDIATE copies (a request into a slot some wave still reads would show as a wrong answer) and with the
    LATEST completion the kernel's s_waitcnt counts allow (a count that is too generous leaves a slab or an X tile stale); bit for
    bit against the one-tile-per-group kernel (KMX_PW_V2=0) and the two emulated convolution launches."""
    code = PW2_CODE
    k2 = {"KMX_PW_KERNEL": "2"}  # (the default is pointwise3_kernel.h since round 5: its test follows)
    envs = (dict(k2, KMX_PW_GRID="1"), dict(k2, KMX_PW_GRID="2"), {"KMX_PW_V2": "0"},
            dict(k2, KMX_PW_GRID="1", KMX_EMU_LATE_DMA="1"), dict(k2, KMX_PW_GRID="2", KMX_EMU_LATE_DMA="2"))  # ... and with the latest legal completion (at the wait / at the barrier after it)
    runs = run_parallel([([sys.executable, "-c", code, emu_full_lib], dict(os.environ, **env)) for env in envs])
    for env, (rc, so, se) in zip(envs, runs):
        assert rc == 0 and "RESULT " in so, (so + se)[-3000:]
        r = json.loads(so.split("RESULT ")[1])
        print(env, r)
        assert all(r["same"]) and r["off_board_zero"], (env, r)
        for e, sc, k in zip(r["err"], r["scale"], (1, 4, 4)):
            assert e <= 2 * 2.0 ** -7 * max(sc, 1.0) * k, (env, r)


def test_resident_weights_seam_kernel_emulated(emu_full_lib):
    """pointwise3_kernel.h (round 5: W1 and W2 resident on the CU, one wave per SIMD, 64-cell tiles) on the CPU: a work-group that walks
    all six tiles of the case (five full, a tail of 18 cells; KMX_PW_GRID=1), two and three work-groups that share them - X double
    buffering, the prefetch one tile ahead, the once-only fetch of the shared W2 rows, the A2 image reuse across tiles - with IMMEDIATE
    copies and with the LATEST completion its one s_waitcnt count allows (at the wait / at the barrier after it); bit for bit against
    the round-3 persistent kernel, the one-tile-per-group kernel and the two emulated convolution launches."""
    envs = ({"KMX_PW_KERNEL": "3", "KMX_PW_GRID": "1"}, {"KMX_PW_KERNEL": "3", "KMX_PW_GRID": "2"}, {"KMX_PW_KERNEL": "3", "KMX_PW_GRID": "3", "KMX_EMU_LATE_DMA": "1"},
            {"KMX_PW_KERNEL": "3", "KMX_PW_GRID": "1", "KMX_EMU_LATE_DMA": "2"}, {"KMX_PW_KERNEL": "2", "KMX_PW_GRID": "1"}, {"KMX_PW_KERNEL": "1"})
    runs = run_parallel([([sys.executable, "-c", PW2_CODE, emu_full_lib], dict(os.environ, **env)) for env in envs])
    digests = []
    for env, (rc, so, se) in zip(envs, runs):
        assert rc == 0 and "RESULT " in so, (env, (so + se)[-3000:])
        r = json.loads(so.split("RESULT ")[1])
        print(env, r)
        assert all(r["same"]) and r["off_board_zero"], (env, r)
        for e, sc, k in zip(r["err"], r["scale"], (1, 4, 4)):
            assert e <= 2 * 2.0 ** -7 * max(sc, 1.0) * k, (env, r)
        digests.append(r["digest"])
    assert len(set(digests)) == 1, digests  # the three kernels agree bit for bit
