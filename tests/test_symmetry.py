"""Symmetry maps (copyWithSymmetry, cpp/neuralnet/nninputs.cpp:529-597): the oracle's restatement against numpy
flips/transposes, the inverse property, and equivariance of the whole oracle net under input symmetries."""
import numpy as np
import pytest

from conftest import make_rows
from oracle import oracle


def np_sym(img, s):  # bit0 flipY, bit1 flipX, bit2 transpose, applied in that order
    if s & 1:
        img = img[::-1]
    if s & 2:
        img = img[:, ::-1]
    if s & 4 and img.shape[0] == img.shape[1]:
        img = img.transpose(1, 0, 2)
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("h,w", [(19, 19), (9, 9), (7, 13), (13, 7)])
def test_symmetry_matches_numpy(h, w):
    rng = np.random.default_rng(h * 100 + w)
    img = rng.standard_normal((h, w, 3)).astype(np.float32)
    for s in range(8):
        fwd = oracle.copyWithSymmetry(img, s, False)
        assert np.array_equal(fwd, np_sym(img, s)), s
        back = oracle.copyWithSymmetry(fwd, s, True)
        assert np.array_equal(back, img), s  # outputs use the inverse map (reverse=true, :595-597)


def test_net_outputs_invariant_to_eval_symmetry(small_model):
    """Evaluating with symmetry s and un-symmetrising the outputs is the same function of the position as s=0
    only for a symmetric net; for a random net it must equal evaluating the pre-symmetrised position with s=0."""
    rng = np.random.default_rng(3)
    sp, gl = make_rows(rng, 1)
    m = oracle.loadModelFile(small_model)
    for s in range(8):
        a = oracle.getOutput(m, 19, 19, sp, gl, [s])
        img = oracle.copyWithSymmetry(sp.reshape(19, 19, 22), s, False).reshape(1, 361, 22)
        b = oracle.getOutput(m, 19, 19, img, gl, [0])
        pol_b = oracle.copyWithSymmetry(b["policy"][0, :361].reshape(19, 19, 1), s, True).reshape(361)
        own_b = oracle.copyWithSymmetry(b["ownership"][0].reshape(19, 19, 1), s, True).reshape(361)
        assert np.allclose(a["policy"][0, :361], pol_b, atol=1e-5) and np.allclose(a["ownership"][0], own_b, atol=1e-5)
        assert np.allclose(a["policy"][0, 361], b["policy"][0, 361], atol=1e-5) and np.allclose(a["value"], b["value"], atol=1e-5)
