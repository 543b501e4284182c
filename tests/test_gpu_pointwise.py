"""-m gpu: the fused seam kernel of two 1x1 convolutions (pointwise_kernel.h) — against a numpy restatement, bit for bit
against the two convolution launches it replaces, and inside whole nets (b18c384nbt with KMX_FUSE_SEAMS=1 vs 0)."""
import ctypes
import os

import numpy as np
import pytest

import pointwise_ref as ref
from conftest import make_rows
from katago_amd import capi, modelgen, nninterface as nn

pytestmark = pytest.mark.gpu
C1, C2, C3 = 192, 384, 192  # b18c384nbt: mid -> trunk -> mid


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("batch,L,acts", [(1, 19, (2, 2)), (3, 13, (2, 1)), (2, 9, (1, 0)), (40, 19, (2, 2)), (131, 19, (2, 2)), (256, 19, (2, 2))])
def test_seam_kernel_against_numpy_and_against_two_launches(dtype, batch, L, acts):
    """cells = batch * L * L is in general not a multiple of the 128-cell tile (tail tile), boards carry masks. (mish, mish) on
    19x19 runs the persistent kernel (pointwise2_kernel.h): batch 131 gives 370 tiles - work-groups with two tiles and with one -
    batch 256 the bench shape (722 tiles: three and two per work-group)."""
    rng = np.random.default_rng(batch * 100 + L)
    cells = batch * L * L
    mask = np.ones((batch, L, L), np.float32)
    if batch > 1:
        mask[1, :, L - 3:] = 0  # a narrower board inside the buffer
        mask[-1, L // 2:, :] = 0
    x, resid, w1, s1, b1, w2, s2, b2, m = ref.make_case(rng, cells, C1, C2, C3, mask.reshape(-1))
    fused = nn.testEvaluatePointwisePair(batch, L, L, dtype, x, resid, w1, s1, b1, acts[0], w2, s2, b2, acts[1], m, True)
    plain = nn.testEvaluatePointwisePair(batch, L, L, dtype, x, resid, w1, s1, b1, acts[0], w2, s2, b2, acts[1], m, False)
    for f, p, name in zip(fused, plain, ("trunk_raw", "mid_raw", "mid_act")):
        assert np.array_equal(f, p), "%s differs from the two-launch path (max %g)" % (name, np.abs(f - p).max())
    want = ref.seam(x, resid, w1, s1, b1, acts[0], w2, s2, b2, acts[1], m, dtype)
    # 16-bit rounding of an fp32-accumulated value against the float64 restatement: one unit in the last place of the format
    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
    for f, w, name in zip(fused, want, ("trunk_raw", "mid_raw", "mid_act")):
        err = np.abs(f - w)
        lim = 2 * ulp * np.maximum(np.abs(w), 1.0) * (4 if name != "trunk_raw" else 1)  # the second GEMM sees the first's rounding
        assert np.isfinite(f).all() and (err <= lim).all(), (name, float(err.max()), float(np.abs(w).max()))
    on = m == 1.0
    assert (fused[2][~on] == 0).all()  # activated image is zero off the board


def test_seam_kernel_refuses_unsupported_shapes():
    rng = np.random.default_rng(0)
    x, resid, w1, s1, b1, w2, s2, b2, m = ref.make_case(rng, 81, 64, 96, 64)
    with pytest.raises(nn.KatamxError) as e:
        nn.testEvaluatePointwisePair(1, 9, 9, "bf16", x, resid, w1, s1, b1, 2, w2, s2, b2, 2, m, True)
    assert e.value.code == capi.KMX_ERR_UNSUPPORTED
    out = nn.testEvaluatePointwisePair(1, 9, 9, "bf16", x, resid, w1, s1, b1, 2, w2, s2, b2, 2, m, False)  # the plain path takes any shape
    assert all(np.isfinite(o).all() for o in out)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_whole_net_with_and_without_fused_seams_is_bit_identical(tmp_path, dtype):
    """b18c384nbt: 17 seams between its 18 nested-bottleneck blocks. Same rows through a handle built with KMX_FUSE_SEAMS=0
    and one with the default: all four outputs equal bit for bit, at batch sizes on both sides of the fusion threshold and
    on the two-engine path; the profile shows 17 seam launches per pass and 2 remaining plain 1x1 launches (+ heads)."""
    nn.globalInitialize()
    p = str(tmp_path / "b18.bin")
    modelgen.write_model(p, "b18c384nbt", seed=31)
    ctx = nn.createComputeContext([0], 19, 19, precision=dtype)
    model = nn.loadModelFile(p)
    rng = np.random.default_rng(31)
    sp, gl = make_rows(rng, 256, 19, [(19, 19), (13, 13), (9, 9), (19, 10)] * 64)
    sym = rng.integers(0, 8, 256).astype(np.int32)
    old = os.environ.get("KMX_FUSE_SEAMS")
    try:
        os.environ["KMX_FUSE_SEAMS"] = "0"
        h0 = nn.createComputeHandle(ctx, model, 256)
        os.environ["KMX_FUSE_SEAMS"] = "1"
        h1 = nn.createComputeHandle(ctx, model, 256)
    finally:
        if old is None:
            os.environ.pop("KMX_FUSE_SEAMS", None)
        else:
            os.environ["KMX_FUSE_SEAMS"] = old
    for n in (256, 100, 37, 24, 5):
        a = nn.getOutput(h0, sp[:n], gl[:n], sym[:n])
        b = nn.getOutput(h1, sp[:n], gl[:n], sym[:n])
        for k in a:
            if not np.array_equal(a[k], b[k]):
                d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).reshape(n, -1)
                rows = np.where(~(d == 0).all(axis=1))[0]
                # which of the two is the odd one out? evaluate both again
                a2 = nn.getOutput(h0, sp[:n], gl[:n], sym[:n])
                b2 = nn.getOutput(h1, sp[:n], gl[:n], sym[:n])
                raise AssertionError((n, k, float(np.nanmax(d)), "rows", rows[:20].tolist(), "unfused repeatable", bool(np.array_equal(a[k], a2[k])),
                                      "fused repeatable", bool(np.array_equal(b[k], b2[k])), "second pass equal", bool(np.array_equal(a2[k], b2[k]))))
    lib = h1._lib

    def profile(h):
        capi.check(lib.kmx_handle_set_split_min(h._p, 0), lib)
        capi.check(lib.kmx_handle_set_profiling(h._p, 1), lib)
        nn.getOutput(h, sp, gl, sym)
        ent = (capi.ProfileEntry * 32)()
        cnt = ctypes.c_int()
        capi.check(lib.kmx_handle_get_profile(h._p, ent, 32, ctypes.byref(cnt)), lib)
        return {ent[i].name.decode(): (int(ent[i].launches), round(ent[i].total_ms, 3)) for i in range(cnt.value)}

    p0, p1 = profile(h0), profile(h1)
    print("unfused", p0)
    print("fused  ", p1)
    assert "conv1x1_pair" not in p0 and p1["conv1x1_pair"][0] == 17
    assert p0["conv3x3"][0] == p1["conv3x3"][0] == 73 and p0["conv1x1"][0] == p1["conv1x1"][0] + 34
    assert p1["conv1x1_pair"][1] + p1["conv1x1"][1] < p0["conv1x1"][1]  # and the 1x1 work takes less time
    h0.close()
    h1.close()
