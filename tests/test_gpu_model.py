"""-m gpu: the whole hot path through the C ABI (kmx_eval == NeuralNet::getOutput) against the oracle and the
committed reference-torch golden vectors; size-independent properties at the BASELINE batch size."""
import os

import numpy as np
import pytest

import parity_stats as ps
from conftest import REPO, make_rows
from katago_amd import capi, modelgen, nninterface as nn
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(REPO, "tests", "golden")
_ORACLE_CACHE = {}


def oracle_outputs(key, path, sp, gl, sym, opt=None):
    """The oracle is dtype-independent: compute it once per case, not once per device precision."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.getOutput(oracle.loadModelFile(path), 19, 19, sp, gl, sym, opt)
    return _ORACLE_CACHE[key]


def outputs_close(got, want, mask, rel, ab):
    n = mask.shape[0]
    full = np.concatenate([mask, np.ones((n, 1), bool)], axis=1)
    ok = True
    for name, g, w in (("policy", got["policy"][full], want["policy"][full]), ("value", got["value"], want["value"]),
                       ("score", got["score"], want["score"]), ("ownership", got["ownership"][mask], want["ownership"][mask])):
        err = np.abs(g.astype(np.float64) - w)
        lim = ab + rel * np.maximum(np.abs(g), np.abs(w))
        print("%-9s max err %.4g (scale %.3g)" % (name, err.max(), np.abs(w).max()))
        ok = ok and bool(np.isfinite(g).all() and (err <= lim).all())
    return ok


@pytest.fixture(scope="module")
def ctx19():
    nn.globalInitialize()
    c = {d: nn.createComputeContext([0], 19, 19, precision=d) for d in ("bf16", "fp16", "fp32")}
    yield c


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_reference_torch_golden(ctx19, dtype):
    """HIP path vs the outputs of the reference PyTorch model (tools/gen_torch_golden.py): 13x9 and 9x9 boards in a
    19x19 buffer exercise the mask path; optimism 0/1 select policy channel 0/1."""
    v = np.load(os.path.join(GOLD, "torch_nbt_vectors.npz"))
    h = nn.createComputeHandle(ctx19[dtype], nn.loadModelFile(os.path.join(GOLD, "torch_nbt.bin.gz")), 8)
    mask = v["spatial_nhwc"][:, :, 0] > 0
    for opt in (0.0, 1.0):
        got = nn.getOutput(h, v["spatial_nhwc"], v["glob"], None, np.full(4, opt, np.float32))
        want = dict(policy=v["policy"][:, int(opt), :], value=v["value"], score=v["score"], ownership=v["ownership"])
        assert outputs_close(got, want, mask, 0.03, 0.08 if dtype == "bf16" else 0.02)
    h.close()


# fp32 (round 5): KMX_PREC_FP32 / useFP16Mode = False (nninterface.h:50-63), the verification mode - the same schedule on four-byte tensors,
# one plain launch per convolution (conv_f32.hip): against the fp32 oracle at 1e-4 of the value (summation order, the hardware's exp and
# reciprocal), the deep net at 3e-4
@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("arch,version,act,stem", [("b3c64nbt", 15, "mish", 3), ("b6c96", 11, "relu", 5), ("b10c128", 14, "mish", 3),
                                                   ("b2c32nbt", 9, "relu", 3), ("b2c32nbt", 16, "silu", 3), ("b18c384nbt", 15, "mish", 3)])
def test_model_vs_oracle(ctx19, model_dir, dtype, arch, version, act, stem):
    p = os.path.join(model_dir, "m_%s_v%d.bin.gz" % (arch, version))
    if not os.path.exists(p):
        modelgen.write_model(p, arch, version=version, activation=act, stem_kernel=stem, seed=version)
    rng = np.random.default_rng(version)
    n = 6
    sizes = [(19, 19), (19, 19), (13, 13), (9, 9), (19, 10), (7, 11)]
    sp, gl = make_rows(rng, n, 19, sizes)
    sym = np.array([0, 5, 3, 6, 7, 1], np.int32)
    opt = np.array([0, 0.3, 1.0, 0.0, 0.5, 0.2], np.float32)
    want = oracle_outputs(("model", arch, version), p, sp, gl, sym, opt)
    h = nn.createComputeHandle(ctx19[dtype], nn.loadModelFile(p), 16)
    assert h.precision == dtype
    got = nn.getOutput(h, sp, gl, sym, opt)
    mask = sp[:, :, 0] > 0
    deep = arch == "b18c384nbt"
    rel, ab = (0.05, 0.15 if deep else 0.08) if dtype == "bf16" else (0.02, 0.05 if deep else 0.02)
    if dtype == "fp32":
        rel, ab = (3e-4, 3e-4) if deep else (1e-4, 1e-4)
    assert outputs_close(got, want, mask, rel, ab)
    # rows are independent: a batch of 1 reproduces row 0 bit for bit, whatever else shares the batch
    one = nn.getOutput(h, sp[:1], gl[:1], sym[:1], opt[:1])
    assert np.array_equal(one["policy"][0], got["policy"][0]) and np.array_equal(one["value"][0], got["value"][0])
    # counters (nneval.cpp:712-713): rows, batches
    assert h.stats() == (n + 1, 2)
    # includeOwnerMap=false rows skip the ownership copy
    noown = nn.getOutput(h, sp[:2], gl[:2], sym[:2], opt[:2], includeOwnerMap=False)
    assert noown["ownership"] is None and np.array_equal(noown["policy"], got["policy"][:2])
    with pytest.raises(nn.KatamxError):
        nn.getOutput(h, np.repeat(sp, 3, 0), np.repeat(gl, 3, 0))  # 18 rows > maxBatchSize 16
    h.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_error_statistics_reference_thresholds(ctx19, model_dir, dtype):
    """The reference's cross-backend acceptance test (testnnevalcanary.cpp:573-829) restated: 99th percentile and max
    error of winrate / lead / score / top policy / policy KL over a batch of positions, against its thresholds for
    reduced-precision backends (:806-807)."""
    results = {}
    for arch, npos in (("b10c128", 192), ("b18c384nbt", 48)):
        p = os.path.join(model_dir, "stat_%s.bin.gz" % arch)
        if not os.path.exists(p):
            modelgen.write_model(p, arch, seed=77)
        rng = np.random.default_rng(99)
        sizes = [(19, 19)] * (npos * 3 // 4) + [(13, 13)] * (npos // 8) + [(9, 9)] * (npos - npos * 3 // 4 - npos // 8)
        sp, gl = make_rows(rng, npos, 19, sizes)
        sym = rng.integers(0, 8, npos).astype(np.int32)
        model = nn.loadModelFile(p)
        info = nn.getModelDesc(model)["postProcessParams"]
        h = nn.createComputeHandle(ctx19[dtype], model, 64)
        got = {k: [] for k in ("policy", "value", "score", "ownership")}
        for i in range(0, npos, 64):  # several batches, ragged tail
            o = nn.getOutput(h, sp[i:i + 64], gl[i:i + 64], sym[i:i + 64])
            for k in got:
                got[k].append(o[k])
        got = {k: np.concatenate(v) for k, v in got.items()}
        want = oracle_outputs(("stats", arch), p, sp, gl, sym)
        stats = ps.error_stats(ps.postprocess(want, sp, info), ps.postprocess(got, sp, info))
        print(arch, dtype, {k: float("%.4g" % v) for k, v in stats.items()})
        results[arch] = ps.check(stats, ps.LIMITS_REDUCED)
        h.close()
    assert not any(results.values()), results


@pytest.mark.parametrize("arch", ["b28c512nbt", "b40c256"])
def test_error_statistics_large_nets_default_precision(ctx19, model_dir, arch):
    """BASELINE configs[3]'s nets under the reference's cross-backend acceptance statistics (as above), in the backend's DEFAULT
    precision (fp16 with the 1/8 range transform): 99th percentile and max of winrate / lead / score / top-policy / policy-KL
    errors against the fp32 oracle within the reference's limits for reduced-precision backends (testnnevalcanary.cpp:806-807)."""
    p = os.path.join(model_dir, "big_%s.bin.gz" % arch)
    if not os.path.exists(p):
        modelgen.write_model(p, arch, seed=28)
    npos = 32
    rng = np.random.default_rng(123)
    sizes = [(19, 19)] * 24 + [(13, 13)] * 4 + [(9, 9)] * 4
    sp, gl = make_rows(rng, npos, 19, sizes)
    sym = rng.integers(0, 8, npos).astype(np.int32)
    model = nn.loadModelFile(p)
    info = nn.getModelDesc(model)["postProcessParams"]
    ctx = nn.createComputeContext([0], 19, 19, precision="auto")
    h = nn.createComputeHandle(ctx, model, 32)
    assert h.precision == "fp16"
    got = nn.getOutput(h, sp, gl, sym)
    want = oracle_outputs(("bigstats", arch), p, sp, gl, sym)
    stats = ps.error_stats(ps.postprocess(want, sp, info), ps.postprocess(got, sp, info))
    print(arch, {k: float("%.4g" % v) for k, v in stats.items()})
    assert not ps.check(stats, ps.LIMITS_REDUCED), stats
    h.close()


@pytest.mark.parametrize("precision", ["auto", "bf16"])
def test_full_batch_properties(ctx19, model_dir, precision):
    """BASELINE size (b18c384nbt, batch 256), in the backend's DEFAULT precision (what bench.py times: fp16 with the 1/8 range
    transform) and in bf16: finite outputs; every row equals the same position evaluated in a small batch (bit exact: rows
    never interact); evaluating with symmetry s equals evaluating the pre-symmetrised board with symmetry 0 and
    un-symmetrising the outputs."""
    p = os.path.join(model_dir, "prop_b18.bin")
    if not os.path.exists(p):
        modelgen.write_model(p, "b18c384nbt", seed=5)
    rng = np.random.default_rng(1)
    base, gbase = make_rows(rng, 8)
    sp, gl = np.tile(base, (32, 1, 1)), np.tile(gbase, (32, 1))
    sym = (np.arange(256) // 8 % 8).astype(np.int32)
    ctx = nn.createComputeContext([0], 19, 19, precision="auto") if precision == "auto" else ctx19[precision]
    h = nn.createComputeHandle(ctx, nn.loadModelFile(p), 256)
    assert h.precision == ("fp16" if precision == "auto" else precision)
    big = nn.getOutput(h, sp, gl, sym)
    assert all(np.isfinite(big[k]).all() for k in big)
    small = nn.getOutput(h, sp[:8], gl[:8], sym[:8])
    for k in big:
        assert np.array_equal(big[k][:8], small[k])
    # 256 and 255 rows take the two-engine path of the handle (two streams, halves of 128 / 128+127), 8 rows the
    # single-engine path with a different work-group shape: all bit-identical row by row
    odd = nn.getOutput(h, sp[:255], gl[:255], sym[:255])
    for k in big:
        assert np.array_equal(big[k][:255], odd[k])
    # device-resident entry (kmx_eval_device), asynchronous then synchronised
    import torch
    d_sp, d_gl = torch.from_numpy(sp).cuda(), torch.from_numpy(gl).cuda()
    d_pol, d_val = torch.zeros((256, 362), device="cuda"), torch.zeros((256, 3), device="cuda")
    d_sc, d_own = torch.zeros((256, 6), device="cuda"), torch.zeros((256, 361), device="cuda")
    torch.cuda.synchronize()
    nn.getOutputDevice(h, d_sp.data_ptr(), d_gl.data_ptr(), sym, None, d_pol.data_ptr(), d_val.data_ptr(), d_sc.data_ptr(),
                       d_own.data_ptr(), sync=False)
    h.sync()
    assert np.array_equal(d_pol.cpu().numpy(), big["policy"]) and np.array_equal(d_val.cpu().numpy(), big["value"])
    assert np.array_equal(d_sc.cpu().numpy(), big["score"]) and np.array_equal(d_own.cpu().numpy(), big["ownership"])
    # row r (symmetry s) vs the explicitly symmetrised board at symmetry 0
    for s in (1, 2, 4, 7):
        r = 8 * s
        img = oracle.copyWithSymmetry(sp[r].reshape(19, 19, 22), s, False).reshape(1, 361, 22)
        o = nn.getOutput(h, img, gl[r:r + 1], [0])
        pol = oracle.copyWithSymmetry(o["policy"][0, :361].reshape(19, 19, 1), s, True).reshape(361)
        assert np.array_equal(pol, big["policy"][r, :361]) and np.array_equal(o["value"][0], big["value"][r])
    h.close()


def test_headline_batch_vs_oracle_default_precision(model_dir):
    """BASELINE configs[1] end to end in the DEFAULT precision against the oracle: b18c384nbt, ONE batch of 256 distinct
    positions (full boards, 13x13, 9x9 and rectangular boards in the 19x19 buffer; every symmetry; optimism 0..1) through the
    handle bench.py times - two engines on two streams, the 8-wave convolution shapes and the persistent seam kernel - and 16
    of its rows through the fp32 oracle (~1 s of CPU): the first and last row of the batch, both sides of the split between the
    two half-batch engines, and small boards from either half. Tolerance as for the deep nets above: 2 % of the value + 0.05
    (the reference's own layer tolerance is 3 % of max(|x|, 3), cpp/tests/testnn.cpp:8-15), and its cross-backend statistics
    (testnnevalcanary.cpp:806-807) over the sample."""
    p = os.path.join(model_dir, "prop_b18.bin")
    if not os.path.exists(p):
        modelgen.write_model(p, "b18c384nbt", seed=5)
    rng = np.random.default_rng(256)
    n = 256
    sizes = ([(19, 19)] * 5 + [(13, 13), (9, 9), (19, 10)]) * (n // 8)
    sp, gl = make_rows(rng, n, 19, sizes)
    sym = rng.integers(0, 8, n).astype(np.int32)
    opt = rng.random(n).astype(np.float32)
    model = nn.loadModelFile(p)
    h = nn.createComputeHandle(nn.createComputeContext([0], 19, 19, precision="auto"), model, 256)
    assert h.precision == "fp16"
    big = nn.getOutput(h, sp, gl, sym, opt)
    assert all(np.isfinite(v).all() for v in big.values())
    rows = np.array([0, 1, 5, 6, 7, 64, 126, 127, 128, 129, 133, 134, 135, 200, 254, 255])
    assert {sizes[r] for r in rows} == {(19, 19), (13, 13), (9, 9), (19, 10)}
    want = oracle_outputs(("headline256",), p, sp[rows], gl[rows], sym[rows], opt[rows])
    got = {k: v[rows] for k, v in big.items()}
    assert outputs_close(got, want, sp[rows][:, :, 0] > 0, 0.02, 0.05)
    info = nn.getModelDesc(model)["postProcessParams"]
    stats = ps.error_stats(ps.postprocess(want, sp[rows], info), ps.postprocess(got, sp[rows], info))
    print("headline batch, default precision:", {k: float("%.4g" % v) for k, v in stats.items()})
    assert not ps.check(stats, ps.LIMITS_REDUCED), stats
    # the same rows in a batch of 16 (one engine, the 4-wave / twelve-wave shapes, two plain convolutions per seam): bit-identical
    part = nn.getOutput(h, sp[rows], gl[rows], sym[rows], opt[rows])
    for k in part:
        assert np.array_equal(part[k], got[k]), k
    h.close()


def test_handle_errors(ctx19, small_model):
    model = nn.loadModelFile(small_model)
    with pytest.raises(nn.KatamxError):
        nn.createComputeHandle(ctx19["bf16"], model, 0)
    # (fp32 was refused until round 5 - KMX_ERR_UNSUPPORTED, asserted here; it is a served mode now, and what must be refused is a
    # precision mode that does not exist)
    nn.createComputeContext([0], 19, 19, precision="fp32").close()
    with pytest.raises(nn.KatamxError) as e:
        nn.createComputeContext([0], 19, 19, precision=99)
    assert e.value.code == capi.KMX_ERR_INVALID_ARG
    with pytest.raises(nn.KatamxError):
        nn.createComputeContext([0], 25, 19)
    h = nn.createComputeHandle(ctx19["bf16"], model, 4)
    sp, gl = make_rows(np.random.default_rng(0), 2)
    with pytest.raises(nn.KatamxError):
        nn.getOutput(h, sp, gl, [0, 9])  # symmetry out of range
    h.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_metadata_encoder_net(ctx19, model_dir, dtype):
    """Nets with an sgf-metadata encoder (a19; desc.cpp:1571-1625, eigenbackend.cpp:1848-1860,1929-1932): the HIP path
    against the reference-torch golden vectors and, on a bigger random net and a batch that takes the two-engine path,
    against the oracle; the metadata input is mandatory for such nets and forbidden for the others."""
    v = np.load(os.path.join(GOLD, "torch_meta_vectors.npz"))
    model = nn.loadModelFile(os.path.join(GOLD, "torch_meta.bin.gz"))
    assert model.info.meta_encoder_version == 1 and model.info.num_input_meta_channels == 192
    h = nn.createComputeHandle(ctx19[dtype], model, 8)
    mask = v["spatial_nhwc"][:, :, 0] > 0
    n = mask.shape[0]
    got = nn.getOutput(h, v["spatial_nhwc"], v["glob"], None, np.zeros(n, np.float32), rowMeta=v["meta"])
    want = dict(policy=v["policy"][:, 0, :], value=v["value"], score=v["score"], ownership=v["ownership"])
    assert outputs_close(got, want, mask, 0.03, 0.08 if dtype == "bf16" else 0.02)
    with pytest.raises(nn.KatamxError):
        nn.getOutput(h, v["spatial_nhwc"], v["glob"])  # metadata missing
    h.close()

    p = os.path.join(model_dir, "meta_b3c64nbt.bin.gz")
    if not os.path.exists(p):
        modelgen.write_model(p, "b3c64nbt", seed=77, version=15, meta_encoder=48)
    rng = np.random.default_rng(77)
    n = 230
    sp, gl = make_rows(rng, n)
    meta = (rng.random((n, 192)) < 0.15).astype(np.float32)
    sym = rng.integers(0, 8, n).astype(np.int32)
    want = oracle_outputs(("meta", n), p, sp, gl, sym) if False else oracle.getOutput(oracle.loadModelFile(p), 19, 19, sp, gl, sym, None, True, 0, meta)
    h = nn.createComputeHandle(ctx19[dtype], nn.loadModelFile(p), 256)
    got = nn.getOutput(h, sp, gl, sym, None, rowMeta=meta)
    assert outputs_close(got, want, sp[:, :, 0] > 0, 0.05 if dtype == "bf16" else 0.02, 0.1 if dtype == "bf16" else 0.03)
    h.close()
    # a net WITHOUT an encoder refuses a metadata input (the reference asserts the same pairing)
    p2 = os.path.join(model_dir, "plain_b2c32nbt.bin.gz")
    if not os.path.exists(p2):
        modelgen.write_model(p2, "b2c32nbt", seed=1)
    h2 = nn.createComputeHandle(ctx19[dtype], nn.loadModelFile(p2), 8)
    with pytest.raises(nn.KatamxError):
        nn.getOutput(h2, sp[:2], gl[:2], None, None, rowMeta=meta[:2])
    h2.close()


def test_packed_input_rows_bit_exact(ctx19, model_dir):
    """the bit-packed input entry (kmx_eval_packed; the reference's binaryInputNCHWPacked layout) gives
    bit-identical outputs to the fp32 rows — small batch, and a 256-row batch through the two-engine path."""
    p = os.path.join(model_dir, "packed_b3c64nbt.bin.gz")
    if not os.path.exists(p):
        modelgen.write_model(p, "b3c64nbt", seed=21)
    rng = np.random.default_rng(21)
    sizes = [(19, 19), (13, 13), (9, 9), (19, 10)] * 64
    sp, gl = make_rows(rng, 256, 19, sizes)
    sym = rng.integers(0, 8, 256).astype(np.int32)
    opt = rng.random(256).astype(np.float32)
    pk = nn.packRows(sp, 19, 19)
    assert pk.shape == (256, 22 * 46)
    h = nn.createComputeHandle(ctx19["bf16"], nn.loadModelFile(p), 256)
    for n in (5, 256):
        a = nn.getOutput(h, sp[:n], gl[:n], sym[:n], opt[:n])
        b = nn.getOutputPacked(h, pk[:n], gl[:n], sym[:n], opt[:n])
        for k in a:
            assert np.array_equal(a[k], b[k]), (n, k)
    h.close()


@pytest.mark.parametrize("precision,rel,floor", [
    # 52 (b28) / 80 (b40) convolutions deep. fp16 with the 1/8 range transform (the backend default since round 3): 2 % of the value plus
    # 0.08; bf16 - the precision configs[3] NAMES - has 8 mantissa bits instead of 11: 5 % + 0.4 (the limits round 2 measured it under)
    ("fp16", 0.02, 0.08),
    ("bf16", 0.05, 0.4),
])
@pytest.mark.parametrize("arch", ["b28c512nbt", "b40c256"])
def test_large_nets_of_the_analysis_config(ctx19, model_dir, arch, precision, rel, floor):
    """BASELINE configs[3]: b28c512nbt / b40c256 (random weights), bf16 AND the default fp16, batch 512 (cpp/configs/analysis_example.cfg:95-132).
    (a) 8 rows - full and small boards, symmetries - against the oracle at the 16-bit tolerance of the deep nets, written per precision;
    (b) a 512-row batch on a handle created for 512 rows (split over two engines, the 8-wave x 128/192-channel convolution
        shapes at full batch) reproduces, bit for bit, the same rows evaluated 16 at a time; counters add up."""
    p = os.path.join(model_dir, "big_%s.bin.gz" % arch)
    if not os.path.exists(p):
        modelgen.write_model(p, arch, seed=28)
    rng = np.random.default_rng(28)
    n = 512
    sizes = ([(19, 19)] * 5 + [(13, 13), (9, 9), (19, 10)]) * (n // 8)
    sp, gl = make_rows(rng, n, 19, sizes)
    sym = rng.integers(0, 8, n).astype(np.int32)
    opt = rng.random(n).astype(np.float32)
    model = nn.loadModelFile(p)
    h = nn.createComputeHandle(ctx19[precision], model, 512)
    assert h.precision == precision
    big = nn.getOutput(h, sp, gl, sym, opt)
    want = oracle_outputs(("big", arch), p, sp[:8], gl[:8], sym[:8], opt[:8])
    small = {k: v[:8] for k, v in big.items()}
    assert outputs_close(small, want, sp[:8, :, 0] > 0, rel, floor)
    for i in (0, 16, 240, 496):
        part = nn.getOutput(h, sp[i:i + 16], gl[i:i + 16], sym[i:i + 16], opt[i:i + 16])
        for k in part:
            assert np.array_equal(part[k], big[k][i:i + 16]), (arch, precision, i, k)
    assert h.stats() == (512 + 64, 5)
    assert all(np.isfinite(v).all() for v in big.values())
    h.close()
