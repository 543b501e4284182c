#!/usr/bin/env python3
"""Generate tests/golden/torch_tf{a,b}.* : golden vectors for transformer trunks from the REFERENCE PyTorch model.

Same recipe as tools/gen_torch_golden.py (runs only where /root/reference exists; imports the reference's
python/katago/train/model_pytorch.py, re-randomises every weight with a fixed seed, exports with the reference's own
python/export_model_pytorch.py, evaluates with torch on CPU) for the version-17 block kinds of SURVEY.md 8 row f4:

  torch_tfa  attnrope / ffnsg trunk (fixed-theta 2D RoPE, 4 heads = 4 KV heads), per-cell RMSNorm trunk tip
  torch_tfb  conv nested-bottleneck block + nested-bottleneck transformer block + attnrope / ffnsg with grouped-query
             attention (4 query heads on 2 KV heads, q/k head dim 8, v head dim 4), learnable RoPE, per-board
             ("spatial") RMSNorm trunk tip

tests/test_oracle_torch.py checks the C oracle against them. Nothing from the reference is copied: this script only
*runs* it.
"""
import gzip
import os
import runpy
import shutil
import sys
import tempfile

import numpy as np

REF = os.environ.get("KATAGO_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REF, "python"))

import torch  # noqa: E402
from katago.train import model_pytorch, modelconfigs  # noqa: E402

BASE = dict(modelconfigs.config_of_name["b7c96h3tfrs"])
BASE.update(trunk_num_channels=32, mid_num_channels=16, gpool_num_channels=8, p1_num_channels=8, g1_num_channels=8,
            v1_num_channels=12, sbv2_num_channels=16, num_scorebeliefs=2, v2_size=16, transformer_ffn_channels=48,
            activation="mish", trunk_final_rmsnorm=True)
CONFIGS = {
    "torch_tfa": dict(BASE, transformer_heads=4, transformer_kv_heads=4,
                      block_kind=[["attn1", "attnrope"], ["ffn1", "ffnsg"], ["attn2", "attnrope"], ["ffn2", "ffnsg"]]),
    "torch_tfb": dict(BASE, transformer_heads=4, transformer_kv_heads=2, attention_query_head_dim=8, attention_value_head_dim=4,
                      learnable_rope=True, trunk_rmsnorm_spatial=True,
                      block_kind=[["rconv1", "bottlenest2"], ["tnest1", "bottlenest2transformerropesg"], ["attn1", "attnrope"],
                                  ["ffn1", "ffnsg"]]),
}
for _name, _cfg in CONFIGS.items():
    modelconfigs.config_of_name["kmxtest-" + _name] = _cfg

_captured = {}
_orig_initialize = model_pytorch.Model.initialize


def _initialize_and_randomise(self):
    _orig_initialize(self)
    g = torch.Generator().manual_seed(20260922)
    with torch.no_grad():
        for pname, p in self.named_parameters():
            if pname.endswith("rope_freqs"):
                p.copy_((torch.rand(p.shape, generator=g) * 2.0 - 1.0) * 0.7)  # rad / square, both signs
            elif p.dim() >= 2 and p.numel() > p.shape[0] and not (p.dim() == 4 and p.shape[0] == 1 and p.shape[2] == 1 and p.shape[3] == 1):
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.2 / fan_in) ** 0.5)
            elif "norm" in pname and pname.endswith("weight"):
                p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.3)  # RMSNorm gains around 1
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    _captured["model"] = self


model_pytorch.Model.initialize = _initialize_and_randomise


def make_inputs():
    # 4 rows on a 19x19 buffer; row 2 is a 13x9 board, row 3 a 9x9 board (attention key mask, per-board RMS count)
    rng = np.random.default_rng(11)
    n, L = 4, 19
    sizes = [(19, 19), (19, 19), (13, 9), (9, 9)]  # (x_size, y_size)
    spatial = np.zeros((n, 22, L, L), dtype=np.float32)
    for b, (xs, ys) in enumerate(sizes):
        spatial[b, 0, :ys, :xs] = 1.0
        stones = rng.random((ys, xs))
        spatial[b, 1, :ys, :xs] = stones < 0.25
        spatial[b, 2, :ys, :xs] = (stones >= 0.25) & (stones < 0.5)
        for c in range(3, 22):
            spatial[b, c, :ys, :xs] = rng.random((ys, xs)) < 0.08
    glob = rng.normal(0.0, 0.5, (n, 19)).astype(np.float32)
    return spatial, glob, sizes


def generate(name):
    tmp = tempfile.mkdtemp(prefix="kmxgolden")
    argv = sys.argv
    sys.argv = ["export_model_pytorch.py", "-export-random-initialized-model", "kmxtest-" + name, "-export-dir", tmp, "-model-name",
                "kmxtest-" + name, "-filename-prefix", "model"]
    try:
        runpy.run_path(os.path.join(REF, "python", "export_model_pytorch.py"), run_name="__main__")
    finally:
        sys.argv = argv
    model = _captured["model"]
    model.eval()
    with open(os.path.join(tmp, "model.bin"), "rb") as f, gzip.open(os.path.join(OUT, name + ".bin.gz"), "wb", 9) as g:
        shutil.copyfileobj(f, g)
    spatial, glob, sizes = make_inputs()
    n, L = spatial.shape[0], spatial.shape[2]
    with torch.no_grad():
        outputs = model(torch.from_numpy(spatial), torch.from_numpy(glob))
    main_head = model.float32ify_output(outputs)[0]
    out_policy, out_value, out_misc, out_moremisc, out_ownership = [t.numpy() for t in main_head[:5]]
    # export channel mapping as in tools/gen_torch_golden.py
    policy = np.stack([out_policy[:, 0, :], out_policy[:, 5, :]], axis=1)
    score = np.concatenate([out_misc[:, 0:4], out_moremisc[:, 0:2]], axis=1)
    np.savez_compressed(
        os.path.join(OUT, name + "_vectors.npz"),
        spatial_nhwc=np.ascontiguousarray(spatial.transpose(0, 2, 3, 1)).reshape(n, L * L, 22),
        glob=glob,
        policy=policy.astype(np.float32),
        value=out_value.astype(np.float32),
        score=score.astype(np.float32),
        ownership=out_ownership.reshape(n, L * L).astype(np.float32),
        sizes=np.array(sizes, dtype=np.int32),
    )
    shutil.rmtree(tmp, ignore_errors=True)
    print("wrote", os.path.join(OUT, name + ".bin.gz"), os.path.join(OUT, name + "_vectors.npz"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for nm in CONFIGS:
        generate(nm)
