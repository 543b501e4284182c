"""numpy restatement of the seam of two 1x1 convolutions (NestedBottleneckResidualBlock::apply at a block boundary,
eigenbackend.cpp:1308-1314; BatchNormLayer::apply :739-762) for the unit tests of pointwise_kernel.h, with the device's
rounding points: 16-bit operands, fp32 accumulation, 16-bit trunk / activated images."""
import numpy as np


def round16(x, dtype):
    import torch

    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(torch.bfloat16 if dtype == "bf16" else torch.float16).float().numpy()


def act(x, kind):
    if kind == 1:
        return np.maximum(x, 0)
    if kind == 2:  # mish, softplus linearised above 20 as the reference does
        sp = np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))
        return x * np.tanh(sp)
    if kind == 3:
        return x / (1 + np.exp(-x))
    return x


def seam(x, resid, w1, s1, b1, a1, w2, s2, b2, a2, mask, dtype):
    """x [cells][c1], resid [cells][c2], w [out][in], mask [cells] -> (trunk_raw, mid_raw, mid_act) as the device stores them"""
    x16, r16, w1_16, w2_16 = (round16(v, dtype) for v in (x, resid, w1, w2))
    trunk = r16.astype(np.float64) + x16.astype(np.float64) @ w1_16.T.astype(np.float64)
    on = (mask == 1.0)[:, None]
    t_act = round16(np.where(on, act(trunk * s1 + b1, a1), 0.0), dtype)
    mid = t_act.astype(np.float64) @ w2_16.T.astype(np.float64)
    mid_act = np.where(on, act(mid * s2 + b2, a2), 0.0)
    return round16(trunk, dtype), round16(mid, dtype), round16(mid_act, dtype)


def make_case(rng, cells, c1, c2, c3, mask=None):
    x = (rng.normal(size=(cells, c1)) * (rng.random((cells, c1)) < 0.6)).astype(np.float32)
    resid = rng.normal(size=(cells, c2)).astype(np.float32)
    w1 = (rng.normal(size=(c2, c1)) / np.sqrt(c1)).astype(np.float32)
    w2 = (rng.normal(size=(c3, c2)) / np.sqrt(c2)).astype(np.float32)
    s1, b1 = rng.uniform(0.5, 1.5, c2).astype(np.float32), rng.normal(0, 0.3, c2).astype(np.float32)
    s2, b2 = rng.uniform(0.5, 1.5, c3).astype(np.float32), rng.normal(0, 0.3, c3).astype(np.float32)
    if mask is None:
        mask = np.ones(cells, np.float32)
    return x, resid, w1, s1, b1, w2, s2, b2, mask.astype(np.float32)
