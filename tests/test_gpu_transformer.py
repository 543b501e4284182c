"""-m gpu, EXPERIMENTAL: model-v17 transformer trunks on the HIP backend (transformer_kernels.hip + the 1x1 convolution
kernel) against the reference PyTorch goldens and the oracle. The device kernels of these layers have not run on
hardware yet; the loader refuses such nets unless KMX_EXPERIMENTAL_TRANSFORMER=1, and these tests only run with it:

    KMX_EXPERIMENTAL_TRANSFORMER=1 python -m pytest tests/test_gpu_transformer.py -m gpu -x -q
"""
import os

import numpy as np
import pytest

from conftest import REPO, make_rows
from katago_amd import nninterface as nn
from oracle import oracle
from test_gpu_model import outputs_close

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("KMX_EXPERIMENTAL_TRANSFORMER") != "1", reason="experimental: set KMX_EXPERIMENTAL_TRANSFORMER=1")]
GOLD = os.path.join(REPO, "tests", "golden")
REF_MODELS = os.path.join(REPO, "oracle", "_ref", "models")


@pytest.fixture(scope="module")
def ctx19():
    nn.globalInitialize()
    return {d: nn.createComputeContext([0], 19, 19, precision=d) for d in ("bf16", "fp16")}


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["torch_tfa", "torch_tfb"])
def test_transformer_torch_golden(ctx19, name, dtype):
    """tools/gen_torch_golden_tf.py: attention + SwiGLU FFN trunks (fixed / learnable RoPE, GQA, nested transformer
    bottleneck, per-cell / per-board RMSNorm tips), 13x9 and 9x9 boards inside the 19x19 buffer."""
    v = np.load(os.path.join(GOLD, name + "_vectors.npz"))
    h = nn.createComputeHandle(ctx19[dtype], nn.loadModelFile(os.path.join(GOLD, name + ".bin.gz")), 8)
    mask = v["spatial_nhwc"][:, :, 0] > 0
    for opt in (0.0, 1.0):
        got = nn.getOutput(h, v["spatial_nhwc"], v["glob"], None, np.full(4, opt, np.float32))
        want = dict(policy=v["policy"][:, int(opt), :], value=v["value"], score=v["score"], ownership=v["ownership"])
        assert outputs_close(got, want, mask, 0.03, 0.08 if dtype == "bf16" else 0.02)
    h.close()


@pytest.mark.parametrize("net", ["b7c96h3tfrs-test5-cnorm.bin.gz", "b7c96h6kv3qk32v16tflrs-fson-bnh.bin.gz"])
def test_reference_transformer_nets_vs_oracle(ctx19, net):
    """The trained transformer test nets of the reference (cpp/rungpuerrortest.sh:32-34), bf16, all 8 symmetries, a batch
    that mixes 19x19, 13x13 and 9x9 boards."""
    p = os.path.join(REF_MODELS, net)
    if not os.path.exists(p):
        pytest.skip("reference test nets not packaged")
    rng = np.random.default_rng(5)
    sp, gl = make_rows(rng, 8, 19, [(19, 19)] * 4 + [(13, 13)] * 2 + [(9, 9)] * 2)
    sym = np.arange(8, dtype=np.int32)
    want = oracle.getOutput(oracle.loadModelFile(p), 19, 19, sp, gl, sym)
    h = nn.createComputeHandle(ctx19["bf16"], nn.loadModelFile(p), 8)
    got = nn.getOutput(h, sp, gl, sym)
    assert outputs_close(got, want, sp[:, :, 0] > 0, 0.05, 0.15)
    # rows are independent: the same row alone gives the same answer
    one = nn.getOutput(h, sp[5:6], gl[5:6], sym[5:6])
    for k in ("policy", "value", "score", "ownership"):
        assert np.array_equal(one[k][0], got[k][5])
    h.close()
