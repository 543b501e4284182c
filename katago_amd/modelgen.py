"""Synthetic KataGo model files (seeded random weights) in the reference's .bin format.

There is no network on the build/bench machines, so the benchmark and the parity tests run on
random-initialised nets of the exact published architectures. The file format is the one parsed by
the reference's cpp/neuralnet/desc.cpp (header :2441-2570, trunk :1669-1768, blocks :566-818,
policy head :2051-2104, value head :2242-2273) and written by python/export_model_pytorch.py:117-330;
architectures follow python/katago/train/modelconfigs.py (b18c384nbt :605-641, b28c512nbt :876-920,
b40c256 :545, b6c96, b10c128).

Weights are scaled so that activations stay O(1) through the whole trunk (He-style fan-in scaling, a
damping factor on every residual branch), and every batch-norm gets a non-trivial scale and bias, so that
a numerical comparison exercises every layer.
"""
import gzip
import struct

import numpy as np

ARCHS = {
    # name: (trunk C, mid, gpool, block kinds, p1, g1, v1, v2, activation)
    "b18c384nbt": dict(C=384, mid=192, gpool=64, blocks=["n", "n", "ng"] * 5 + ["n", "n", "n"], p1=48, g1=48, v1=96, v2=128),
    "b28c512nbt": dict(C=512, mid=256, gpool=64, blocks=(["n", "n", "ng"] * 9 + ["n"]), p1=64, g1=64, v1=128, v2=144),
    "b40c256": dict(C=256, mid=256, gpool=64,
                    blocks=[("g" if i % 5 == 4 and i < 39 else "r") for i in range(40)], p1=48, g1=48, v1=64, v2=96),
    "b10c128": dict(C=128, mid=128, gpool=32, blocks=["r", "r", "g", "r", "r", "g", "r", "r", "g", "r"], p1=32, g1=32, v1=32, v2=64),
    "b6c96": dict(C=96, mid=96, gpool=32, blocks=["r", "r", "g", "r", "g", "r"], p1=32, g1=32, v1=32, v2=48),
    # small nets for fast CPU tests (same block structure as the nbt family)
    "b2c32nbt": dict(C=32, mid=16, gpool=8, blocks=["n", "ng"], p1=8, g1=8, v1=12, v2=16),
    "b3c64nbt": dict(C=64, mid=32, gpool=16, blocks=["n", "ng", "n"], p1=12, g1=12, v1=24, v2=32),
}

ACT_NAMES = {"relu": "ACTIVATION_RELU", "mish": "ACTIVATION_MISH", "silu": "ACTIVATION_SILU", "identity": "ACTIVATION_IDENTITY"}


class _Writer:
    def __init__(self, f, rng, act, text=False, version=15):
        self.version = version
        self.f = f
        self.rng = rng
        self.act = act
        self.text = text

    def ln(self, s):
        self.f.write((str(s) + "\n").encode("ascii"))

    def floats(self, arr):
        arr = np.ascontiguousarray(arr, dtype="<f4").reshape(-1)
        if self.text:  # .txt models: whitespace-separated decimal tokens (desc.cpp:44-51); %.9g round-trips fp32
            self.f.write((" ".join("%.9g" % v for v in arr) + "\n").encode("ascii"))
            return
        self.f.write(b"@BIN@")
        self.f.write(arr.tobytes())
        self.f.write(b"\n")

    def conv(self, name, k, cin, cout, gain=1.0):
        self.ln(name)
        for v in (k, k, cin, cout, 1, 1):
            self.ln(v)
        std = gain * np.sqrt(2.0 / (k * k * cin))
        self.floats(self.rng.standard_normal((k, k, cin, cout)) * std)  # file order y,x,ic,oc

    def bn(self, name, c):
        self.ln(name)
        self.ln(c)
        self.ln(1e-20)
        self.ln(1)
        self.ln(1)
        self.floats(np.zeros(c))  # mean
        self.floats(np.full(c, 1.0 - 1e-20))  # variance
        self.floats(self.rng.uniform(0.6, 1.4, c) * self.rng.choice([1.0, 1.0, 1.0, -1.0], c))  # scale
        self.floats(self.rng.normal(0.0, 0.25, c))  # bias

    def activation(self, name, kind=None):
        self.ln(name)
        if self.version >= 11:  # older formats have no activation-kind token: relu is implied (desc.cpp:384-402)
            self.ln(ACT_NAMES[kind or self.act])

    def matmul(self, name, cin, cout, gain=1.0):
        self.ln(name)
        self.ln(cin)
        self.ln(cout)
        self.floats(self.rng.standard_normal((cin, cout)) * (gain / np.sqrt(cin)))

    def matbias(self, name, c):
        self.ln(name)
        self.ln(c)
        self.floats(self.rng.normal(0.0, 0.2, c))


def _ordinary(w, name, c, mid):
    w.ln("ordinary_block")
    w.ln(name)
    w.bn(name + ".norm1", c)
    w.activation(name + ".act1")
    w.conv(name + ".conv1", 3, c, mid)
    w.bn(name + ".norm2", mid)
    w.activation(name + ".act2")
    w.conv(name + ".conv2", 3, mid, c, gain=0.35)


def _gpool(w, name, c, regular, gpool):
    w.ln("gpool_block")
    w.ln(name)
    w.bn(name + ".norm1", c)
    w.activation(name + ".act1")
    w.conv(name + ".conv1r", 3, c, regular)
    w.conv(name + ".conv1g", 3, c, gpool)
    w.bn(name + ".normg", gpool)
    w.activation(name + ".actg")
    w.matmul(name + ".linear_g", 3 * gpool, regular, gain=0.5)
    w.bn(name + ".norm2", regular)
    w.activation(name + ".act2")
    w.conv(name + ".conv2", 3, regular, c, gain=0.35)


def _nested(w, name, c, mid, gpool, with_gpool):
    w.ln("nested_bottleneck_block")
    w.ln(name)
    w.ln(2)
    w.bn(name + ".normp", c)
    w.activation(name + ".actp")
    w.conv(name + ".convp", 1, c, mid)
    if with_gpool:
        _gpool(w, name + ".blockstack.0", mid, mid - gpool, gpool)
    else:
        _ordinary(w, name + ".blockstack.0", mid, mid)
    _ordinary(w, name + ".blockstack.1", mid, mid)
    w.bn(name + ".normq", mid)
    w.activation(name + ".actq")
    w.conv(name + ".convq", 1, mid, c, gain=0.35)


def write_model(path, arch, seed=20260921, version=15, activation="mish", name=None, stem_kernel=3, meta_encoder=None, stem_gain=1.0):
    """Write a random-weight model file. `path` may end in .bin, .bin.gz, .txt or .txt.gz. Returns the architecture dict.
    meta_encoder = internal channel count of an sgf-metadata encoder (export_model_pytorch.py:493-504; version >= 15), or None.
    stem_gain multiplies the stem's weights: the layers behind it preserve magnitudes, so every tensor of the net grows with it
    (tests of the fp16 range transform use it to push activations past 65504)."""
    if meta_encoder and version < 15:
        raise ValueError("the sgf-metadata encoder needs model version >= 15")
    a = ARCHS[arch] if isinstance(arch, str) else arch
    rng = np.random.default_rng(seed)
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as f:
        w = _Writer(f, rng, activation, text=path.endswith(".txt") or path.endswith(".txt.gz"), version=version)
        w.ln(name or ("kmxrand-" + (arch if isinstance(arch, str) else "custom")))
        w.ln(version)
        w.ln(22)
        w.ln(19)
        if version >= 13:
            for v in (20.0, 20.0, 20.0, 20.0, 40.0, 0.25, 150.0):
                w.ln(v)
        if version >= 15:
            w.ln(1 if meta_encoder else 0)  # metaEncoderVersion
            for _ in range(7):
                w.ln(0)
        C, mid, gp = a["C"], a["mid"], a["gpool"]
        w.ln("trunk")
        for v in (len(a["blocks"]), C, mid, mid - gp, gp, gp):
            w.ln(v)
        if version >= 15:
            for _ in range(6):
                w.ln(0)
        w.conv("model.conv_spatial", stem_kernel, 22, C, gain=stem_gain)
        w.matmul("model.linear_global", 19, C, gain=0.5 * stem_gain)
        if meta_encoder:
            e = "model.sgf_metadata_encoder"
            w.ln(e)
            w.ln(192)
            w.matmul(e + ".mul1", 192, meta_encoder, gain=1.5)
            w.matbias(e + ".bias1", meta_encoder)
            w.activation(e + ".act1")
            w.matmul(e + ".mul2", meta_encoder, meta_encoder, gain=1.5)
            w.matbias(e + ".bias2", meta_encoder)
            w.activation(e + ".act2")
            w.matmul(e + ".mul3", meta_encoder, C, gain=0.7)
        for i, kind in enumerate(a["blocks"]):
            bname = "model.blocks.%d" % i
            if kind == "r":
                _ordinary(w, bname, C, mid)
            elif kind == "g":
                _gpool(w, bname, C, mid - gp, gp)
            elif kind in ("n", "ng"):
                _nested(w, bname, C, mid, gp, kind == "ng")
            else:
                raise ValueError(kind)
        w.bn("model.norm_trunkfinal", C)
        w.activation("model.act_trunkfinal")
        # policy head
        npol = 4 if version == 16 else (2 if version >= 12 else 1)
        w.ln("policyhead")
        if version >= 17:
            w.ln(npol)
            for _ in range(3):
                w.ln(0)
        w.conv("policyhead.conv1p", 1, C, a["p1"])
        w.conv("policyhead.conv1g", 1, C, a["g1"])
        w.bn("policyhead.biasg", a["g1"])
        w.activation("policyhead.actg")
        w.matmul("policyhead.linear_g", 3 * a["g1"], a["p1"], gain=0.5)
        w.bn("policyhead.bias2", a["p1"])
        w.activation("policyhead.act2")
        w.conv("policyhead.conv2p", 1, a["p1"], npol)
        if version >= 15:
            w.matmul("policyhead.linear_pass", 3 * a["g1"], a["p1"])
            w.matbias("policyhead.linear_pass.bias", a["p1"])
            w.activation("policyhead.act_pass")
            w.matmul("policyhead.linear_pass2", a["p1"], npol)
        else:
            w.matmul("policyhead.linear_pass", 3 * a["g1"], npol)
        # value head
        w.ln("valuehead")
        if version >= 17:
            for _ in range(3):
                w.ln(0)
        w.conv("valuehead.conv1", 1, C, a["v1"])
        w.bn("valuehead.bias1", a["v1"])
        w.activation("valuehead.act1")
        w.matmul("valuehead.linear2", 3 * a["v1"], a["v2"])
        w.matbias("valuehead.bias2", a["v2"])
        w.activation("valuehead.act2")
        w.matmul("valuehead.linear_valuehead", a["v2"], 3)
        w.matbias("valuehead.bias_valuehead", 3)
        nsv = 6 if version >= 9 else 4
        w.matmul("valuehead.linear_miscvaluehead", a["v2"], nsv)
        w.matbias("valuehead.bias_miscvaluehead", nsv)
        w.conv("valuehead.conv_ownership", 1, a["v1"], 1)
    return a


def mac_per_position(arch, stem_kernel=3, version=15):
    """Direct-convolution multiply-accumulates per board point (SURVEY.md 8d / BASELINE.md section 2)."""
    a = ARCHS[arch] if isinstance(arch, str) else arch
    C, mid, gp = a["C"], a["mid"], a["gpool"]

    def ordinary(c, m):
        return 9 * c * m + 9 * m * c

    def gpoolb(c, r, g):
        return 9 * c * r + 9 * c * g + 9 * r * c

    total = stem_kernel * stem_kernel * 22 * C
    for kind in a["blocks"]:
        if kind == "r":
            total += ordinary(C, mid)
        elif kind == "g":
            total += gpoolb(C, mid - gp, gp)
        else:
            total += C * mid + mid * C + ordinary(mid, mid)
            total += gpoolb(mid, mid - gp, gp) if kind == "ng" else ordinary(mid, mid)
    npol = 4 if version == 16 else (2 if version >= 12 else 1)
    total += C * a["p1"] + C * a["g1"] + a["p1"] * npol + C * a["v1"] + a["v1"]
    return total
