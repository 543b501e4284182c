"""Runs whole nets through the CPU-emulated build of libkatamx (tests/fakehip/emulate_engine.cpp) and prints, as JSON, the
largest deviation of each output from the reference values (PyTorch goldens or the oracle). Own process: it replaces the
library handle of katago_amd.capi, which must not leak into other tests.
    python run_emulated_nets.py <libkatamx_emu.so> <case> [<case> ...]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from katago_amd import capi  # noqa: E402

capi._lib = capi.load_library(path=sys.argv[1])
from conftest import make_rows  # noqa: E402
from katago_amd import modelgen, nninterface as nn  # noqa: E402
from oracle import oracle  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
REF_MODELS = os.path.join(REPO, "oracle", "_ref", "models")


def deviations(got, want, mask):
    n = mask.shape[0]
    full = np.concatenate([mask, np.ones((n, 1), bool)], axis=1)
    return {
        "policy": [float(np.abs(got["policy"] - want["policy"])[full].max()), float(np.abs(want["policy"][full]).max())],
        "value": [float(np.abs(got["value"] - want["value"]).max()), float(np.abs(want["value"]).max())],
        "score": [float(np.abs(got["score"] - want["score"]).max()), float(np.abs(want["score"]).max())],
        "ownership": [float(np.abs(got["ownership"] - want["ownership"])[mask].max()), float(np.abs(want["ownership"][mask]).max())],
        "finite": bool(all(np.isfinite(got[k]).all() for k in got)),
    }


def golden_case(ctx, name, with_meta=False, rows=None):
    v = np.load(os.path.join(GOLD, name + "_vectors.npz"))
    h = nn.createComputeHandle(ctx, nn.loadModelFile(os.path.join(GOLD, name + ".bin.gz")), 8)
    sel = slice(None) if rows is None else rows  # e.g. [2]: only the 13x9 board of the fixture
    n = v["glob"][sel].shape[0]
    got = nn.getOutput(h, v["spatial_nhwc"][sel], v["glob"][sel], None, np.zeros(n, np.float32), rowMeta=v["meta"][sel] if with_meta else None)
    want = dict(policy=v["policy"][sel][:, 0, :], value=v["value"][sel], score=v["score"][sel], ownership=v["ownership"][sel])
    h.close()
    return deviations(got, want, v["spatial_nhwc"][sel][:, :, 0] > 0)


def oracle_case(ctx, path, n, sizes, seed):
    rng = np.random.default_rng(seed)
    sp, gl = make_rows(rng, n, 19, sizes)
    sym = (np.arange(n) * 3 % 8).astype(np.int32)
    opt = np.linspace(0.0, 1.0, n).astype(np.float32)
    want = oracle.getOutput(oracle.loadModelFile(path), 19, 19, sp, gl, sym, opt)
    h = nn.createComputeHandle(ctx, nn.loadModelFile(path), n)
    got = nn.getOutput(h, sp, gl, sym, opt)
    h.close()
    return deviations(got, want, sp[:, :, 0] > 0)


def api_variants(ctx):
    """The entry points that must give the SAME bits as kmx_eval: bit-packed rows, a handle that splits the batch over two
    engines (KMX_SPLIT_MIN forces it at 4 rows here), rows evaluated alone, the counters."""
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_variants.bin")
    modelgen.write_model(p, "b2c32nbt", seed=5)
    rng = np.random.default_rng(9)
    n = 7
    sp, gl = make_rows(rng, n, 19, [(19, 19), (13, 9), (9, 9), (19, 19), (7, 7), (19, 19), (9, 13)])
    sym = (np.arange(n) % 8).astype(np.int32)
    opt = np.linspace(0.0, 1.0, n).astype(np.float32)
    model = nn.loadModelFile(p)
    h = nn.createComputeHandle(ctx, model, n)
    base = nn.getOutput(h, sp, gl, sym, opt)
    packed = nn.getOutputPacked(h, nn.packRows(sp, 19, 19), gl, sym, opt)
    alone = nn.getOutput(h, sp[4:5], gl[4:5], sym[4:5], opt[4:5])
    rows, batches = h.stats()
    h.close()
    os.environ["KMX_SPLIT_MIN"] = "4"
    h2 = nn.createComputeHandle(ctx, model, n)
    del os.environ["KMX_SPLIT_MIN"]
    split = nn.getOutput(h2, sp, gl, sym, opt)
    small = nn.getOutput(h2, sp[:3], gl[:3], sym[:3], opt[:3])  # below the threshold: one engine
    rows2, batches2 = h2.stats()
    h2.close()
    keys = ("policy", "value", "score", "ownership")
    return {
        "packed_equal": bool(all(np.array_equal(base[k], packed[k]) for k in keys)),
        "alone_equal": bool(all(np.array_equal(base[k][4], alone[k][0]) for k in keys)),
        "split_equal": bool(all(np.array_equal(base[k], split[k]) for k in keys)),
        "small_equal": bool(all(np.array_equal(base[k][:3], small[k]) for k in keys)),
        "stats": [int(rows), int(batches), int(rows2), int(batches2)],
        "finite": bool(all(np.isfinite(base[k]).all() for k in keys)),
    }


def batcher_case(ctx):
    """The persistent leaf batcher (kmx_batcher_*): rows submitted from several threads come back bit-identical to kmx_eval on the
    same rows, whatever batch they land in; counters; a non-binary plane is refused by submit with KMX_ERR_INVALID_ARG."""
    import threading
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_batcher.bin")
    modelgen.write_model(p, "b2c32nbt", seed=6)
    rng = np.random.default_rng(10)
    n = 10
    sp, gl = make_rows(rng, n, 19, [(19, 19), (13, 9), (9, 9), (19, 19), (7, 7)] * 2)
    sym = (np.arange(n) % 8).astype(np.int32)
    opt = np.linspace(0.0, 1.0, n).astype(np.float32)
    model = nn.loadModelFile(p)
    h = nn.createComputeHandle(ctx, model, n)
    base = nn.getOutput(h, sp, gl, sym, opt)
    h.close()
    b = nn.Batcher(ctx, model, 4, maxInFlight=2)
    got = [None] * n

    packed = nn.packRows(sp, 19, 19)  # odd rows go in as bit planes (kmx_batcher_submit_packed), even ones as fp32 rows

    def worker(i0):
        for i in range(i0, n, 3):
            t = b.submit(packed[i], gl[i], sym[i], opt[i], True, packed=True) if i % 2 else b.submit(sp[i], gl[i], sym[i], opt[i], True)
            got[i] = b.wait(t)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    rows, batches = b.stats()
    keys = ("policy", "value", "score", "ownership")
    equal = bool(all(np.array_equal(base[k][i], got[i][k]) for k in keys for i in range(n)))
    bad = sp[0].copy()
    bad[5, 3] = 0.5
    err = ""
    try:
        b.submit(bad, gl[0], 0, 0.0, True)  # refused at once: no row reserved, nobody else's batch fails
    except Exception as e:  # KatamxError
        err = str(e)
    t = b.submit(sp[1], gl[1], sym[1], opt[1], False)  # the batcher keeps working after a failed batch
    again = b.wait(t)
    b.close()
    # more tickets outstanding than the staging sets can hold (3 sets x 2 rows < 10 rows): every thread submits ALL its rows
    # before it waits for the first, as a server thread of the reference's NNEvaluator does with the batch it popped
    b2 = nn.Batcher(ctx, model, 2, maxInFlight=1)
    got2 = [None] * n

    def greedy(i0):
        tickets = [(i, b2.submit(sp[i], gl[i], sym[i], opt[i], True)) for i in range(i0, n, 2)]
        for i, t in tickets:
            got2[i] = b2.wait(t)

    th = [threading.Thread(target=greedy, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    hung = any(t.is_alive() for t in th)
    equal2 = (not hung) and bool(all(np.array_equal(base[k][i], got2[i][k]) for k in keys for i in range(n)))
    if not hung:
        b2.close()
    # a granule below the batch size with growth allowed (KMX_BATCH_QUANTUM=2, KMX_BATCH_GROW_AHEAD=3; batches of up to 6, three in
    # flight; the default seals at one granule always): a batch is sealed at 2 or 4 rows while a slot is free (fewer than three
    # launched or queued), at 6 otherwise - every row must come back the same whichever batch it rode in
    os.environ["KMX_BATCH_QUANTUM"] = "2"
    os.environ["KMX_BATCH_GROW_AHEAD"] = "3"
    b3 = nn.Batcher(ctx, model, 6, maxInFlight=3)
    del os.environ["KMX_BATCH_QUANTUM"], os.environ["KMX_BATCH_GROW_AHEAD"]
    tickets = [b3.submit(sp[i], gl[i], sym[i], opt[i], True) for i in range(n)]
    got3 = [b3.wait(t) for t in tickets]
    rows3, batches3 = b3.stats()
    b3.close()
    equal3 = bool(all(np.array_equal(base[k][i], got3[i][k]) for k in keys for i in range(n)))
    return {"equal": equal, "rows": int(rows), "batches": int(batches), "error": err, "many_tickets_equal": equal2,
            "granule_equal": equal3, "granule_stats": [int(rows3), int(batches3)],
            "after_error_equal": bool(all(np.array_equal(base[k][1], again[k]) for k in ("policy", "value", "score")))}


def main():
    nn.globalInitialize()
    out = {}
    for case in sys.argv[2:]:
        dtype, what = case.split(":")
        ctx = nn.createComputeContext([0], 19, 19, precision=dtype)
        if what == "api_variants":
            out[case] = api_variants(ctx)
            continue
        if what == "batcher":
            out[case] = batcher_case(ctx)
            continue
        if what.split("@")[0] in ("torch_nbt", "torch_tfa", "torch_tfb"):
            name, _, rows = what.partition("@")  # torch_tfa@2 = row 2 only
            out[case] = golden_case(ctx, name, rows=[int(r) for r in rows.split(",")] if rows else None)
        elif what == "torch_meta":
            out[case] = golden_case(ctx, what, with_meta=True)
        elif what == "big_activations":
            # Every tensor of this net is ~2e4 times the usual size (the stem is, the rest preserves magnitudes): trunk values pass
            # 65504. With the fp16 range transform (the default) the device tracks the fp32 oracle; with the file's own values
            # (KMX_FP16_SCALE8=0) fp16 overflows - which is what the transform is for (desc.cpp:2718-2736).
            p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_big.bin")
            modelgen.write_model(p, "b3c64nbt", seed=12, stem_gain=2.0e4)
            out[case] = oracle_case(ctx, p, 2, [(19, 19), (9, 13)], 3)
            os.environ["KMX_FP16_SCALE8"] = "0"
            out[case]["finite_without_transform"] = oracle_case(ctx, p, 2, [(19, 19), (9, 13)], 3)["finite"]
            del os.environ["KMX_FP16_SCALE8"]
        elif what.startswith("gen_"):  # gen_<arch>_v<version>
            _, arch, ver = what.split("_")
            p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_%s_%s.bin" % (arch, ver))
            modelgen.write_model(p, arch, seed=11, version=int(ver[1:]))
            out[case] = oracle_case(ctx, p, 3, [(19, 19), (13, 9), (9, 9)], 1)
        else:  # a file under oracle/_ref/models
            out[case] = oracle_case(ctx, os.path.join(REF_MODELS, what), 1, [(13, 13)], 2)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
