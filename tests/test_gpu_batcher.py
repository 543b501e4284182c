"""-m gpu: the persistent leaf batcher behind the C ABI (kmx_batcher_*, SURVEY 8 row a4 / north_star) on the MI355X.
  * rows submitted concurrently from many threads come back bit-identical to kmx_eval on the same rows, whatever batch they land in;
  * the reference's own `benchmark` (its search, its NNEvaluator, its server threads) on katago_hip_refeval with katamxBatcher = true:
    several server threads feed one batcher; the rate is recorded next to the plain handle's (profiles/)."""
import os
import re
import threading

import numpy as np
import pytest

from conftest import REPO, make_rows
from katago_amd import modelgen
from katago_amd import nninterface as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_batcher_rows_are_bit_identical_to_kmx_eval(tmp_path, dtype):
    nn.globalInitialize()
    p = str(tmp_path / "net.bin")
    modelgen.write_model(p, "b3c64nbt", seed=12)
    ctx = nn.createComputeContext([0], 19, 19, precision=dtype)
    model = nn.loadModelFile(p)
    rng = np.random.default_rng(12)
    n = 600
    sp, gl = make_rows(rng, n, 19, [(19, 19), (13, 13), (9, 9), (19, 10), (7, 11)] * (n // 5))
    sym = rng.integers(0, 8, n).astype(np.int32)
    opt = rng.random(n).astype(np.float32)
    h = nn.createComputeHandle(ctx, model, 200)
    base = {k: np.concatenate([nn.getOutput(h, sp[i:i + 200], gl[i:i + 200], sym[i:i + 200], opt[i:i + 200])[k] for i in range(0, n, 200)])
            for k in ("policy", "value", "score", "ownership")}
    h.close()
    b = nn.Batcher(ctx, model, 48, maxInFlight=3)
    assert b.effectiveBatch() == 48
    big = nn.Batcher(ctx, model, 1024, maxInFlight=1)  # sealed at one granule of the device (its CU count), include/katamx.h ABI 6 / 7
    assert big.effectiveBatch() == min(1024, 256)
    big.close()
    got = [None] * n
    errors = []
    packed = nn.packRows(sp, 19, 19)  # every fifth row is handed over as bit planes (kmx_batcher_submit_packed)

    def worker(k, nthreads):
        try:
            # a few rows in flight per thread, as a search thread with several pending leaves would have
            pending = []
            for i in range(k, n, nthreads):
                if i % 5 == 1:
                    pending.append((i, b.submit(packed[i], gl[i], sym[i], opt[i], i % 3 != 0, packed=True)))
                else:
                    pending.append((i, b.submit(sp[i], gl[i], sym[i], opt[i], i % 3 != 0)))
                if len(pending) >= 4:
                    j, t = pending.pop(0)
                    got[j] = b.wait(t)
            for j, t in pending:
                got[j] = b.wait(t)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k, 12)) for k in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    rows, batches = b.stats()
    assert rows == n and n / 48 <= batches <= n
    for i in range(n):
        for k in ("policy", "value", "score"):
            assert np.array_equal(base[k][i], got[i][k]), (i, k)
        if i % 3 != 0:
            assert np.array_equal(base["ownership"][i], got[i]["ownership"]), i
    # stale ticket
    with pytest.raises(Exception):
        b.wait(12345678901234)
    b.close()


def test_reference_benchmark_through_the_batcher(tmp_path):
    """BASELINE configs[1] through the reference's `benchmark` with the batcher in the binding's getOutput
    (integration/katamxbackend.cpp, katamxBatcher = true): 4 server threads of the reference's NNEvaluator feed ONE batcher
    (2 batches in flight). Recorded beside the plain-handle numbers; the search side (one blocked OS thread per in-flight
    leaf) bounds both."""
    import test_gpu_reference_harness as h

    model = str(tmp_path / "b18.bin.gz")
    modelgen.write_model(model, "b18c384nbt", seed=7)
    lines = []
    rates = {}
    for name, extra, threads in (("handle, 1 server thread", "numNNServerThreadsPerModel = 1\n", "256"),
                                 ("handle, 2 server threads", "numNNServerThreadsPerModel = 2\n", "256,512"),
                                 ("batcher, 4 server threads", "numNNServerThreadsPerModel = 4\nkatamxBatcher = true\nkatamxBatcherInFlight = 2\n", "256,512,1024")):
        cfg = tmp_path / ("bench_%d.cfg" % len(lines))
        cfg.write_text(h.BENCH_CFG + "nnMaxBatchSize = 256\n" + extra)
        rc, out = h.run("benchmark", "-model", model, "-config", str(cfg), "-v", "8000", "-t", threads, "-boardsize", "19", "-n", "3", timeout=900, binary="katago_hip_refeval")
        assert rc == 0, out[-3000:]
        for l in out.replace("\r", "\n").splitlines():
            m = re.search(r"numSearchThreads = +(\d+):.*nnEvals/s = ([\d.]+).*avgBatchSize = ([\d.]+)", l)
            if m:
                rates[(name, int(m.group(1)))] = float(m.group(2))
                lines.append("%s | %s" % (name, l.strip()))
    assert any(k[0].startswith("batcher") for k in rates) and all(v > 0 for v in rates.values()), rates
    # recorded, and guarded only against a collapse: both paths are bounded by the reference's search side
    assert rates[("batcher, 4 server threads", 256)] >= 0.5 * rates[("handle, 1 server thread", 256)], rates
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "reference_benchmark_batcher.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
