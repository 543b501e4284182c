"""Pins the oracle against the REFERENCE PYTORCH MODEL (python/katago/train/model_pytorch.py) on a net with the
block structure of b18c384nbt (nested bottleneck + gpool inner block, mish, v15 heads). The vectors were produced
by tools/gen_torch_golden.py, which runs the reference model and the reference exporter."""
import os

import numpy as np

from conftest import REPO
from oracle import oracle

GOLD = os.path.join(REPO, "tests", "golden")


def test_oracle_matches_reference_torch_model():
    v = np.load(os.path.join(GOLD, "torch_nbt_vectors.npz"))
    m = oracle.loadModelFile(os.path.join(GOLD, "torch_nbt.bin.gz"))
    assert m.info.model_version == 15 and m.info.num_policy_channels == 2
    mask = v["spatial_nhwc"][:, :, 0] > 0
    assert mask.sum(axis=1).tolist() == [361, 361, 117, 81]  # rows 2,3: 13x9 and 9x9 boards in the 19x19 buffer
    full = np.concatenate([mask, np.ones((4, 1), bool)], axis=1)
    for opt in (0.0, 1.0):
        o = oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"], None, np.full(4, opt, np.float32))
        want = v["policy"][:, int(opt), :]
        assert np.abs(o["policy"] - want)[full].max() < 2e-5  # fp32 both sides, different summation order
        assert np.abs(o["value"] - v["value"]).max() < 1e-5
        assert np.abs(o["score"] - v["score"]).max() < 1e-5
        assert np.abs(o["ownership"] - v["ownership"])[mask].max() < 2e-5
    # optimism blends channel 0 and 1 linearly (eigenbackend.cpp:2553-2562)
    o = oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"], None, np.full(4, 0.25, np.float32))
    want = v["policy"][:, 0, :] + 0.25 * (v["policy"][:, 1, :] - v["policy"][:, 0, :])
    assert np.abs(o["policy"] - want)[full].max() < 2e-5


def test_oracle_matches_reference_torch_model_with_metadata_encoder():
    """A net WITH an sgf-metadata encoder ("humanSL" nets; model header metaEncoderVersion 1): vectors from
    tools/gen_torch_golden_meta.py (reference MetadataEncoder, model_pytorch.py:2881-2933, reference exporter)."""
    v = np.load(os.path.join(GOLD, "torch_meta_vectors.npz"))
    m = oracle.loadModelFile(os.path.join(GOLD, "torch_meta.bin.gz"))
    assert m.info.meta_encoder_version == 1 and m.info.num_input_meta_channels == 192
    mask = v["spatial_nhwc"][:, :, 0] > 0
    n = mask.shape[0]
    full = np.concatenate([mask, np.ones((n, 1), bool)], axis=1)
    o = oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"], None, np.zeros(n, np.float32), rowMeta=v["meta"])
    assert np.abs(o["policy"] - v["policy"][:, 0, :])[full].max() < 2e-5
    assert np.abs(o["value"] - v["value"]).max() < 1e-5
    assert np.abs(o["score"] - v["score"]).max() < 1e-5
    assert np.abs(o["ownership"] - v["ownership"])[mask].max() < 2e-5
    # the encoder matters: other metadata -> other outputs; and it is mandatory for such a net
    o2 = oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"], None, np.zeros(n, np.float32), rowMeta=np.zeros_like(v["meta"]))
    assert np.abs(o2["value"] - o["value"]).max() > 1e-3
    try:
        oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"])
        raise AssertionError("a metadata net must reject rows without metadata")
    except oracle.OracleError:
        pass


def _check_transformer_golden(name):
    v = np.load(os.path.join(GOLD, name + "_vectors.npz"))
    m = oracle.loadModelFile(os.path.join(GOLD, name + ".bin.gz"))
    assert m.info.model_version == 17
    mask = v["spatial_nhwc"][:, :, 0] > 0
    assert mask.sum(axis=1).tolist() == [361, 361, 117, 81]
    full = np.concatenate([mask, np.ones((4, 1), bool)], axis=1)
    for opt in (0.0, 1.0):
        o = oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"], None, np.full(4, opt, np.float32))
        assert np.abs(o["policy"] - v["policy"][:, int(opt), :])[full].max() < 2e-5
        assert np.abs(o["value"] - v["value"]).max() < 1e-5
        assert np.abs(o["score"] - v["score"]).max() < 1e-5
        assert np.abs(o["ownership"] - v["ownership"])[mask].max() < 2e-5
    return m, v


def test_oracle_matches_reference_torch_transformer_trunk():
    """Version-17 attention / SwiGLU-FFN trunk with fixed-theta 2D RoPE and a per-cell RMSNorm trunk tip
    (model_pytorch.py TransformerAttentionBlock / TransformerFFNBlock / RMSNormMask; vectors: tools/gen_torch_golden_tf.py)."""
    m, _ = _check_transformer_golden("torch_tfa")
    assert m.info.num_blocks == 4


def test_oracle_matches_reference_torch_transformer_gqa_learnable_rope():
    """Grouped-query attention (4 query heads on 2 KV heads, q/k dim 8, v dim 4), learnable RoPE frequencies, a nested
    bottleneck block whose inner stack is attention+FFN, next to a convolutional nested bottleneck block; per-board
    ("spatial") RMSNorm trunk tip, which has to count only on-board cells on the 13x9 and 9x9 rows."""
    _check_transformer_golden("torch_tfb")


def test_transformer_rows_are_independent_and_translation_of_the_buffer_is_not_assumed():
    """Size-independent properties of the attention path: a row's outputs do not depend on the other rows of the batch,
    and a 9x9 board gives the same answer in a 9x9 buffer as in the corner of a 19x19 buffer (masked keys contribute
    nothing; RoPE angles depend on (x, y), not on the buffer stride)."""
    v = np.load(os.path.join(GOLD, "torch_tfb_vectors.npz"))
    m = oracle.loadModelFile(os.path.join(GOLD, "torch_tfb.bin.gz"))
    z = np.zeros(4, np.float32)
    o = oracle.getOutput(m, 19, 19, v["spatial_nhwc"], v["glob"], None, z)
    o3 = oracle.getOutput(m, 19, 19, v["spatial_nhwc"][3:4], v["glob"][3:4], None, z[:1])
    for k in ("policy", "value", "score", "ownership"):
        assert np.array_equal(o[k][3], o3[k][0])
    small = v["spatial_nhwc"][3].reshape(19, 19, 22)[:9, :9, :].reshape(1, 81, 22).copy()
    o9 = oracle.getOutput(m, 9, 9, small, v["glob"][3:4], None, z[:1])
    idx = (np.arange(9)[:, None] * 19 + np.arange(9)[None, :]).reshape(-1)
    assert np.abs(o9["policy"][0, :81] - o3["policy"][0, idx]).max() < 2e-5
    assert abs(o9["policy"][0, 81] - o3["policy"][0, 361]) < 2e-5
    assert np.abs(o9["value"] - o3["value"]).max() < 1e-5
    assert np.abs(o9["ownership"][0] - o3["ownership"][0, idx]).max() < 2e-5


def test_reference_transformer_test_nets_play_the_star_points():
    """The two trained transformer nets the reference ships for its own GPU tests (cpp/rungpuerrortest.sh:32-34: RoPE +
    RMSNorm tip; GQA + learnable RoPE + BatchNorm tip) run through the oracle: on an empty 19x19 board the four
    strongest policy moves are the four 4-4 points — a weak but real check that trained weights, not only the random
    ones of the fixtures above, are interpreted the way they were trained."""
    import pytest

    d = os.path.join(os.environ.get("KATAGO_REFERENCE", "/root/reference"), "cpp", "tests", "models")
    names = ("b7c96h3tfrs-test5-cnorm.bin.gz", "b7c96h6kv3qk32v16tflrs-fson-bnh.bin.gz")
    if not all(os.path.exists(os.path.join(d, f)) for f in names):
        pytest.skip("reference checkout not present")
    sp = np.zeros((1, 361, 22), np.float32)
    sp[:, :, 0] = 1.0
    gl = np.zeros((1, 19), np.float32)
    gl[:, 5] = 7.5 / 20.0
    for f in names:
        m = oracle.loadModelFile(os.path.join(d, f))
        assert m.info.model_version == 17 and m.info.trunk_num_channels == 96 and m.info.num_blocks == 14
        o = oracle.getOutput(m, 19, 19, sp, gl)
        assert all(np.isfinite(o[k]).all() for k in o)
        assert sorted(np.argsort(-o["policy"][0])[:4].tolist()) == [3 * 19 + 3, 3 * 19 + 15, 15 * 19 + 3, 15 * 19 + 15]
