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
