"""-m gpu: the small-batch 3x3 shapes with their weights in registers (conv_small_kernel.h REGW: cfg 127 - a board's cell tiles over three
work-groups -, 128, 126 - a board x 64 channels) against round 4's slab-ring shapes (KMX_CONV_TUNE=regw=0: cfg 117 / 118 / 119) on the
MI355X: the same MFMAs per output in the same K order and the same epilogue, so the SAME BITS in every output of a whole net - at batch
sizes that take each of the three shapes for the net's 64-channel inner layers, on a 19x19 buffer with smaller boards inside and on a 2x3
buffer (most cell tiles off the board: the case the MI355X found in cfg 126's first version). The shape choice is read once per process,
so each setting runs in its own interpreter."""
import hashlib  # noqa: F401
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CODE = r"""
import sys, json, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from katago_amd import nninterface as nn
from oracle import oracle
from conftest import make_rows
nn.globalInitialize()
out = {}
model = sys.argv[1]
for X, Y, sizes in ((19, 19, [(19, 19), (13, 13), (9, 9), (19, 19)]), (2, 3, [(2, 3), (2, 2), (2, 3), (2, 3)])):
    L = max(X, Y)
    ctx = nn.createComputeContext([0], X, Y, precision="fp16")
    h = nn.createComputeHandle(ctx, nn.loadModelFile(model), 200)
    for n in (8, 64, 200):  # 64-channel inner layers: batch x 2 tiles x 3 <= 256 -> cfg 127; batch x 2 <= 256 -> 128; batch <= 256 -> 126
        rng = np.random.default_rng(100 + n)
        sp_full, gl = make_rows(rng, n, L, (sizes * (n // 4 + 1))[:n])
        sp = np.ascontiguousarray(sp_full.reshape(n, L, L, 22)[:, :Y, :X, :]).reshape(n, X * Y, 22)
        sym = (np.arange(n) %% 8).astype(np.int32)
        got = nn.getOutput(h, sp, gl, sym)
        assert all(np.isfinite(v).all() for v in got.values())
        out["%%dx%%d n%%d" %% (X, Y, n)] = hashlib.sha1(b"".join(np.ascontiguousarray(got[k]).tobytes() for k in sorted(got))).hexdigest()
        if n == 8 and X == 19:  # and the answers are the net's: against the oracle at the fp16 tolerance of the whole-net tests
            want = oracle.getOutput(oracle.loadModelFile(model), X, Y, sp, gl, sym, np.zeros(n, np.float32))
            err = float(np.abs(got["value"] - want["value"]).max())
            assert err <= 0.03 * max(1.0, float(np.abs(want["value"]).max())), err
    h.close()
print("RESULT " + json.dumps(out))
""" % (REPO, os.path.join(REPO, "tests"))


def test_register_weights_shapes_give_the_bits_of_the_slab_ring_shapes(tmp_path):
    from katago_amd import modelgen

    modelgen.ARCHS["b3c128nbt"] = dict(C=128, mid=64, gpool=16, blocks=["n", "ng", "n"], p1=16, g1=16, v1=24, v2=32)
    model = str(tmp_path / "b3c128nbt.bin.gz")
    modelgen.write_model(model, "b3c128nbt", seed=11)
    res = {}
    for tune in ("regw=0", "regw=3"):
        p = subprocess.run([sys.executable, "-c", CODE, model], capture_output=True, text=True, timeout=600, env=dict(os.environ, KMX_CONV_TUNE=tune))
        assert p.returncode == 0 and "RESULT " in p.stdout, (tune, (p.stdout + p.stderr)[-3000:])
        res[tune] = json.loads(p.stdout.split("RESULT ")[1])
    assert res["regw=0"] == res["regw=3"], res
