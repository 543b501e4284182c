"""Model loader of the product (katago_amd/csrc/model_desc.cpp) against the oracle's independent loader, and
its error behaviour (the reference throws StringError: cpp/neuralnet/desc.cpp:2441-2615,2753-2815)."""
import gzip
import hashlib
import os

import pytest

from katago_amd import capi, modelgen, nninterface as nn
from oracle import oracle

REF_MODELS = os.path.join(os.environ.get("KATAGO_REFERENCE", "/root/reference"), "cpp", "tests", "models")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("arch,version,act", [("b3c64nbt", 15, "mish"), ("b6c96", 11, "relu"), ("b2c32nbt", 17, "silu"),
                                             ("b10c128", 14, "mish"), ("b2c32nbt", 9, "relu"), ("b2c32nbt", 16, "mish")])
def test_loaders_agree(model_dir, arch, version, act):
    p = os.path.join(model_dir, "ld_%s_v%d.bin.gz" % (arch, version))
    modelgen.write_model(p, arch, version=version, activation=act)
    a = nn.getModelDesc(nn.loadModelFile(p))
    b = oracle.loadModelFile(p).info
    assert a["modelVersion"] == b.model_version == version
    assert a["numParameters"] == b.num_parameters
    assert a["flopsPerPosition"] == b.flops_per_position == 2.0 * modelgen.mac_per_position(arch, version=version)
    assert a["numPolicyChannels"] == b.num_policy_channels == (4 if version == 16 else 2 if version >= 12 else 1)
    assert a["numScoreValueChannels"] == b.num_score_value_channels == 6
    assert a["postProcessParams"]["outputScaleMultiplier"] == 1.0
    assert a["postProcessParams"]["shorttermScoreErrorMultiplier"] == (150.0 if version >= 13 else 30.0)


def test_b18c384nbt_accounting(model_dir):
    """26 139 072 MAC per board point = 18.87 GFLOP per 19x19 eval."""
    assert modelgen.mac_per_position("b18c384nbt") == 26139072
    assert modelgen.mac_per_position("b28c512nbt") == 72313984 or abs(modelgen.mac_per_position("b28c512nbt") - 72.31e6) < 0.05e6


def test_plain_and_gzip_and_sha256(model_dir):
    p = os.path.join(model_dir, "plain.bin")
    modelgen.write_model(p, "b2c32nbt")
    raw = open(p, "rb").read()
    pz = os.path.join(model_dir, "zipped.bin.gz")
    with gzip.open(pz, "wb") as f:
        f.write(raw)
    a, b = nn.getModelDesc(nn.loadModelFile(p)), nn.getModelDesc(nn.loadModelFile(pz))
    assert a == b
    good = hashlib.sha256(open(pz, "rb").read()).hexdigest()  # digest of the file as stored (fileutils.cpp:117-141)
    nn.loadModelFile(pz, good.upper())
    with pytest.raises(nn.KatamxError) as e:
        nn.loadModelFile(pz, "0" * 64)
    assert e.value.code == capi.KMX_ERR_MODEL and "sha256" in str(e.value)


def test_truncated_and_garbage(model_dir):
    p = os.path.join(model_dir, "full.bin")
    modelgen.write_model(p, "b2c32nbt")
    raw = open(p, "rb").read()
    t = os.path.join(model_dir, "trunc.bin")
    open(t, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(nn.KatamxError) as e:
        nn.loadModelFile(t)
    assert e.value.code == capi.KMX_ERR_MODEL
    g = os.path.join(model_dir, "garbage.bin.gz")
    open(g, "wb").write(b"this is not gzip")
    with pytest.raises(nn.KatamxError) as e:
        nn.loadModelFile(g)
    assert e.value.code == capi.KMX_ERR_IO
    with pytest.raises(nn.KatamxError):
        nn.loadModelFile(os.path.join(model_dir, "missing.bin.gz"))


def test_text_format(model_dir):
    """.txt models carry the floats as text tokens (desc.cpp:44-51): same net, same outputs as the .bin file."""
    import numpy as np
    from conftest import make_rows

    pb, pt = os.path.join(model_dir, "same.bin"), os.path.join(model_dir, "same.txt.gz")
    modelgen.write_model(pb, "b2c32nbt", seed=5)
    modelgen.write_model(pt, "b2c32nbt", seed=5)
    assert nn.getModelDesc(nn.loadModelFile(pb)) == nn.getModelDesc(nn.loadModelFile(pt))
    sp, gl = make_rows(np.random.default_rng(0), 2)
    a = oracle.getOutput(oracle.loadModelFile(pb), 19, 19, sp, gl)
    b = oracle.getOutput(oracle.loadModelFile(pt), 19, 19, sp, gl)
    for k in ("policy", "value", "score", "ownership"):
        assert np.array_equal(a[k], b[k])


def test_reference_nets(model_dir):
    """The real nets shipped with the reference: the v8 convnet loads; transformer nets are rejected cleanly."""
    p = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/models not built")
    a = nn.getModelDesc(nn.loadModelFile(p))
    b = oracle.loadModelFile(p).info
    assert a["name"].startswith("g170-b6c96") and a["numParameters"] == b.num_parameters and a["numPolicyChannels"] == 1
    tf = os.path.join(REF_MODELS, "b7c96h3tfrs-test5-cnorm.bin.gz")
    if os.path.exists(tf):  # a trained model-v17 transformer net of the reference's test set
        t = nn.getModelDesc(nn.loadModelFile(tf))
        assert t["modelVersion"] == 17 and t["numParameters"] == oracle.loadModelFile(tf).info.num_parameters


def _patch_header_token(src, dst, index, value):
    raw = gzip.open(src, "rb").read()
    cut = raw.index(b"@BIN@")
    tokens = raw[:cut].split()
    tokens[index] = value
    with gzip.open(dst, "wb") as f:
        f.write(b" ".join(tokens) + b" " + raw[cut:])
    return tokens


def test_prefer_pass_alive_header_slot(model_dir):
    """tests/testmisc.cpp:159-213 on a v17 file written by the reference's exporter (tests/golden/torch_tfa.bin.gz):
    header = name, version, 22, 19, 7 post-process multipliers, metaEncoderVersion, preferPassAliveUnderSuicideRules,
    6 zeros, 'trunk'; slot values 0 and 1 parse, anything else is refused (desc.cpp:2538-2548). The convolutional v17
    twin goes through the product loader, the transformer file through the oracle's."""
    src = os.path.join(REPO, "tests", "golden", "torch_tfa.bin.gz")
    for val, ok in ((b"0", True), (b"1", True), (b"2", False), (b"-1", False)):
        dst = os.path.join(model_dir, "ppa_tf_%s.bin.gz" % val.decode())
        tokens = _patch_header_token(src, dst, 12, val)
        assert tokens[1] == b"17" and tokens[11] == b"0" and tokens[19] == b"trunk" and len(tokens) > 19
        if ok:
            assert oracle.loadModelFile(dst).info.model_version == 17
        else:
            with pytest.raises(oracle.OracleError):
                oracle.loadModelFile(dst)
    conv = os.path.join(model_dir, "ppa_conv.bin.gz")
    modelgen.write_model(conv, "b2c32nbt", version=17, activation="silu")
    for val, ok in ((b"0", True), (b"1", True), (b"2", False)):
        dst = os.path.join(model_dir, "ppa_conv_%s.bin.gz" % val.decode())
        _patch_header_token(conv, dst, 12, val)
        if ok:
            assert nn.getModelDesc(nn.loadModelFile(dst))["modelVersion"] == 17
            assert oracle.loadModelFile(dst).info.model_version == 17
        else:
            with pytest.raises(nn.KatamxError) as e:
                nn.loadModelFile(dst)
            assert e.value.code == capi.KMX_ERR_MODEL and "preferPassAlive" in str(e.value)


def test_transformer_nets_load_in_the_oracle_and_in_the_backend():
    """Model v17 transformer trunks: both loaders parse them to the same parameter count."""
    for name in ("torch_tfa", "torch_tfb"):
        p = os.path.join(REPO, "tests", "golden", name + ".bin.gz")
        o = oracle.loadModelFile(p).info
        assert o.num_blocks == 4
        m = nn.getModelDesc(nn.loadModelFile(p))
        assert m["modelVersion"] == 17 and m["numParameters"] == o.num_parameters
