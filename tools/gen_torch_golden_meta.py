#!/usr/bin/env python3
"""Generate tests/golden/torch_meta.* : golden vectors for a net WITH an sgf-metadata encoder from the REFERENCE PyTorch
model (python/katago/train/model_pytorch.py: MetadataEncoder :2881-2933, added to the trunk :3743-3745), exported by the
reference's own exporter (export_model_pytorch.py:493-504). Same method as tools/gen_torch_golden.py; runs only where
/root/reference exists. Nothing from the reference is copied: this script only runs it."""
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

NAME = "kmxtest-b2c32nbt-meta"
CONFIG = dict(modelconfigs.config_of_name["b18c384nbt"])
CONFIG.update(
    trunk_num_channels=32, mid_num_channels=16, gpool_num_channels=8,
    block_kind=[["rconv1", "bottlenest2"], ["rconv2", "bottlenest2gpool"]],
    p1_num_channels=8, g1_num_channels=8, v1_num_channels=12, sbv2_num_channels=16, num_scorebeliefs=2, v2_size=16,
    activation="mish",
    metadata_encoder={"meta_encoder_version": 1, "internal_num_channels": 24},
)
modelconfigs.config_of_name[NAME] = CONFIG

_captured = {}
_orig_initialize = model_pytorch.Model.initialize


def _initialize_and_randomise(self):
    _orig_initialize(self)
    g = torch.Generator().manual_seed(20260922)
    with torch.no_grad():
        for pname, p in self.named_parameters():
            if p.dim() >= 2 and p.numel() > p.shape[0] and not (p.dim() == 4 and p.shape[0] == 1 and p.shape[2] == 1 and p.shape[3] == 1):
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.2 / fan_in) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    _captured["model"] = self


model_pytorch.Model.initialize = _initialize_and_randomise


def main():
    tmp = tempfile.mkdtemp(prefix="kmxgoldenmeta")
    argv = sys.argv
    sys.argv = ["export_model_pytorch.py", "-export-random-initialized-model", NAME, "-export-dir", tmp, "-model-name", NAME,
                "-filename-prefix", "model"]
    try:
        runpy.run_path(os.path.join(REF, "python", "export_model_pytorch.py"), run_name="__main__")
    finally:
        sys.argv = argv
    model = _captured["model"]
    model.eval()
    assert model.metadata_encoder is not None
    with open(os.path.join(tmp, "model.bin"), "rb") as f, gzip.open(os.path.join(OUT, "torch_meta.bin.gz"), "wb", 9) as g:
        shutil.copyfileobj(f, g)
    rng = np.random.default_rng(11)
    n, L = 3, 19
    sizes = [(19, 19), (13, 13), (9, 9)]
    spatial = np.zeros((n, 22, L, L), dtype=np.float32)
    for b, (xs, ys) in enumerate(sizes):
        spatial[b, 0, :ys, :xs] = 1.0
        stones = rng.random((ys, xs))
        spatial[b, 1, :ys, :xs] = stones < 0.25
        spatial[b, 2, :ys, :xs] = (stones >= 0.25) & (stones < 0.5)
        for c in range(3, 22):
            spatial[b, c, :ys, :xs] = rng.random((ys, xs)) < 0.08
    glob = rng.normal(0.0, 0.5, (n, 19)).astype(np.float32)
    meta = (rng.random((n, 192)) < 0.2).astype(np.float32) + rng.normal(0.0, 0.3, (n, 192)).astype(np.float32) * (rng.random((n, 192)) < 0.1)
    meta = meta.astype(np.float32)
    with torch.no_grad():
        outputs = model(torch.from_numpy(spatial), torch.from_numpy(glob), torch.from_numpy(meta))
    main_head = model.float32ify_output(outputs)[0]
    out_policy, out_value, out_misc, out_moremisc, out_ownership = [t.numpy() for t in main_head[:5]]
    policy = np.stack([out_policy[:, 0, :], out_policy[:, 5, :]], axis=1)
    score = np.concatenate([out_misc[:, 0:4], out_moremisc[:, 0:2]], axis=1)
    np.savez_compressed(
        os.path.join(OUT, "torch_meta_vectors.npz"),
        spatial_nhwc=np.ascontiguousarray(spatial.transpose(0, 2, 3, 1)).reshape(n, L * L, 22), glob=glob, meta=meta,
        policy=policy.astype(np.float32), value=out_value.astype(np.float32), score=score.astype(np.float32),
        ownership=out_ownership.reshape(n, L * L).astype(np.float32), sizes=np.array(sizes, dtype=np.int32))
    shutil.rmtree(tmp, ignore_errors=True)
    print("wrote", os.path.join(OUT, "torch_meta.bin.gz"), os.path.join(OUT, "torch_meta_vectors.npz"))


if __name__ == "__main__":
    main()
