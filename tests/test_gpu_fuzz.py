"""-m gpu: randomised differential test of the whole path against the oracle — random small architectures (channel
counts that are NOT multiples of the kernels' tile sizes, every block kind, model versions 8-16, 3x3 and 5x5 stems, all
activations), random buffer sizes nnXLen x nnYLen, random real board sizes inside the buffer, random batch sizes,
symmetries and optimism. Seeds are fixed: failures reproduce."""
import os

import numpy as np
import pytest

from conftest import make_rows
from katago_amd import modelgen, nninterface as nn
from oracle import oracle

pytestmark = pytest.mark.gpu
CASES = list(range(48))


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    gp = int(rng.choice([4, 8, 12, 16]))
    mid = gp + int(rng.choice([4, 8, 20, 28, 44]))
    arch = dict(C=int(rng.choice([8, 20, 32, 40, 72, 96])), mid=mid, gpool=gp,
                blocks=[str(rng.choice(["r", "g", "n", "ng"])) for _ in range(int(rng.integers(1, 4)))],
                p1=int(rng.choice([4, 6, 10, 16])), g1=int(rng.choice([3, 8, 12])), v1=int(rng.choice([5, 8, 12, 24])),
                v2=int(rng.choice([7, 16, 40])))
    version = int(rng.choice([8, 9, 10, 11, 12, 13, 14, 15, 16]))
    act = "relu" if version < 11 else str(rng.choice(["relu", "mish", "silu"]))
    stem = int(rng.choice([3, 5]))
    X, Y = int(rng.integers(2, 20)), int(rng.integers(2, 20))
    n = int(rng.choice([1, 2, 3, 7, 16, 33, 230]))  # 230 takes the handle's two-engine path
    return rng, arch, version, act, stem, X, Y, n


@pytest.mark.parametrize("seed", CASES)
def test_random_model_random_board(tmp_path, seed):
    rng, arch, version, act, stem, X, Y, n = random_case(seed)
    path = str(tmp_path / "fuzz.bin.gz")
    modelgen.write_model(path, arch, seed=seed, version=version, activation=act, stem_kernel=stem)
    L = max(X, Y)
    sizes = [(int(rng.integers(2, X + 1)), int(rng.integers(2, Y + 1))) if rng.random() < 0.5 else (X, Y) for _ in range(n)]
    # rows are generated on an LxL canvas and cropped to the X x Y buffer
    sp_full, gl = make_rows(rng, n, L, sizes)
    sp = np.ascontiguousarray(sp_full.reshape(n, L, L, 22)[:, :Y, :X, :]).reshape(n, X * Y, 22)
    sym = rng.integers(0, 8, n).astype(np.int32)
    opt = rng.random(n).astype(np.float32)
    want = oracle.getOutput(oracle.loadModelFile(path), X, Y, sp, gl, sym, opt)
    nn.globalInitialize()
    mask = sp[:, :, 0] > 0
    full = np.concatenate([mask, np.ones((n, 1), bool)], axis=1)
    for dtype, rel, ab in (("bf16", 0.06, 0.12), ("fp16", 0.02, 0.03)):
        ctx = nn.createComputeContext([0], X, Y, precision=dtype)
        h = nn.createComputeHandle(ctx, nn.loadModelFile(path), max(n, 4))
        got = nn.getOutput(h, sp, gl, sym, opt)
        for name, g, w in (("policy", got["policy"][full], want["policy"][full]), ("value", got["value"], want["value"]),
                           ("score", got["score"], want["score"]), ("ownership", got["ownership"][mask], want["ownership"][mask])):
            err = np.abs(g.astype(np.float64) - w)
            scale = max(1.0, float(np.abs(w).max()))
            lim = ab * scale + rel * np.abs(w)
            assert np.isfinite(g).all() and (err <= lim).all(), (
                "seed %d %s %s v%d %s stem%d %dx%d n%d: %s max err %.4g (scale %.3g)" % (seed, dtype, arch, version, act, stem, X, Y, n,
                                                                                          name, err.max(), scale))
        h.close()
