"""-m gpu: EVERY batch size 1 ... 96 of b18c384nbt (the net of BASELINE configs[1] / [2]) on the MI355X, bit for bit against the same rows
evaluated one at a time - 19x19 rows and mixed 9x9 / 13x13 / 19x19 rows in one 19x19 buffer, through kmx_eval (every size exactly) and
through the leaf batcher (kmx_batcher_*: bursts of 1 ... 96 rows, which the dispatcher cuts into whatever batches the device's state
allows, several of them side by side on their own engines - the regime of self-play with a few games per GPU).

Why: the work-group shape of a 3x3 convolution is chosen from the batch size (conv_mfma.hip chooseConvCfg: cfg 125 / 126 / 127 / 128 /
the 4-wave shapes), self-play runs at device batches of 1 ... 64 rows, and round 5's driver run of production self-play died of a GPU
exception while no test had walked these sizes on the net whose channel counts select the shapes. Rows never interact
(nneval.cpp:562-752 serves whatever rows are waiting), so a row's outputs may not depend on the batch it lands in."""
import numpy as np
import pytest

from conftest import make_rows
from katago_amd import modelgen
from katago_amd import nninterface as nn

pytestmark = pytest.mark.gpu

KEYS = ("policy", "value", "score", "ownership")
NMAX = 96


@pytest.fixture(scope="module")
def b18(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("sweep") / "b18c384nbt.bin.gz")
    modelgen.write_model(p, "b18c384nbt", seed=21)
    return p


def _rows(mixed):
    rng = np.random.default_rng(77 + mixed)
    sizes = [(19, 19), (13, 13), (9, 9), (19, 19), (9, 9), (13, 13), (19, 19)] if mixed else [(19, 19)]
    sp, gl = make_rows(rng, NMAX, 19, (sizes * NMAX)[:NMAX])
    sym = rng.integers(0, 8, NMAX).astype(np.int32)
    opt = rng.random(NMAX).astype(np.float32)
    return sp, gl, sym, opt


@pytest.mark.parametrize("mixed", [0, 1], ids=["19x19", "mixed_9_13_19"])
def test_every_batch_size_1_to_96_gives_the_bits_of_batch_1(b18, mixed):
    nn.globalInitialize()
    ctx = nn.createComputeContext([0], 19, 19)
    model = nn.loadModelFile(b18)
    sp, gl, sym, opt = _rows(mixed)
    h = nn.createComputeHandle(ctx, model, NMAX)
    assert h.precision == "fp16"
    one = [nn.getOutput(h, sp[i:i + 1], gl[i:i + 1], sym[i:i + 1], opt[i:i + 1]) for i in range(NMAX)]
    ref = {k: np.concatenate([o[k] for o in one]) for k in KEYS}
    assert all(np.isfinite(ref[k]).all() for k in KEYS)
    bad = []
    for n in range(1, NMAX + 1):
        got = nn.getOutput(h, sp[:n], gl[:n], sym[:n], opt[:n])
        for k in KEYS:
            if not np.array_equal(got[k], ref[k][:n]):
                bad.append((n, k, int(np.argmax(np.any((got[k] != ref[k][:n]).reshape(n, -1), axis=1)))))
    assert not bad, "batch sizes whose rows differ from batch 1 (size, output, first row): %s" % bad[:20]
    # and the tail of a batch: the LAST n rows as a batch (row r of a batch meets other tile / work-group positions than in [:n])
    for n in (7, 15, 21, 22, 43, 64, 85):
        got = nn.getOutput(h, sp[NMAX - n:], gl[NMAX - n:], sym[NMAX - n:], opt[NMAX - n:])
        for k in KEYS:
            assert np.array_equal(got[k], ref[k][NMAX - n:]), (n, k)
    h.close()


@pytest.mark.parametrize("mixed", [0, 1], ids=["19x19", "mixed_9_13_19"])
def test_bursts_of_1_to_96_rows_through_the_batcher_give_the_bits_of_batch_1(b18, mixed):
    nn.globalInitialize()
    ctx = nn.createComputeContext([0], 19, 19)
    model = nn.loadModelFile(b18)
    sp, gl, sym, opt = _rows(mixed)
    h = nn.createComputeHandle(ctx, model, 8)
    one = [nn.getOutput(h, sp[i:i + 1], gl[i:i + 1], sym[i:i + 1], opt[i:i + 1]) for i in range(NMAX)]
    h.close()
    packed = nn.packRows(sp, 19, 19)
    b = nn.Batcher(ctx, model, 64, maxInFlight=3)  # what the binding creates for configs[2] (nnMaxBatchSize 64)
    total = 0
    for n in list(range(1, NMAX + 1)) + [64, 63, 65, 23, 24, 22] * 4:
        start = (n * 7) % NMAX
        idx = [(start + j) % NMAX for j in range(n)]
        tickets = [b.submit(packed[i], gl[i], sym[i], opt[i], j % 4 != 0, packed=True) for j, i in enumerate(idx)]
        for j, (i, t) in enumerate(zip(idx, tickets)):
            got = b.wait(t)
            for k in ("policy", "value", "score"):
                assert np.array_equal(got[k], one[i][k][0]), (n, i, k)
            if j % 4 != 0:
                assert np.array_equal(got["ownership"], one[i]["ownership"][0]), (n, i)
        total += n
    rows, batches = b.stats()
    assert rows == total and batches >= NMAX
    b.close()
