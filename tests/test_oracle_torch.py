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
