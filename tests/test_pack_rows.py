"""Bit-packed spatial rows: kmx_pack_row and nninterface.packRows produce the reference's
binaryInputNCHWPacked layout (dataio/trainingwrite.h:180-183; packBits, dataio/trainingwrite.cpp:314-337): plane by
plane, 8 cells per byte, MOST significant bit first, planes zero-padded to whole bytes. No GPU needed."""
import ctypes

import numpy as np

from conftest import make_rows
from katago_amd import capi, nninterface as nn


def reference_pack_bits(binary_floats):
    """packBits restated literally (trainingwrite.cpp:314-337)."""
    n = len(binary_floats)
    out = np.zeros((n + 7) // 8, np.uint8)
    for i in range(0, n, 8):
        for di in range(min(8, n - i)):
            out[i >> 3] |= np.uint8(int(binary_floats[i + di]) << (7 - di))
    return out


def test_pack_rows_match_reference_layout():
    lib = capi.load_library()
    rng = np.random.default_rng(3)
    for X, Y in ((19, 19), (9, 9), (13, 7), (2, 3)):
        L = max(X, Y)
        sp_full, _ = make_rows(rng, 2, L, [(X, Y), (max(2, X - 1), Y)])
        sp = np.ascontiguousarray(sp_full.reshape(2, L, L, 22)[:, :Y, :X, :]).reshape(2, X * Y, 22)
        pk = nn.packRows(sp, X, Y)
        PB = (X * Y + 7) // 8
        assert pk.shape == (2, 22 * PB)
        for i in range(2):
            want = np.concatenate([reference_pack_bits(sp[i][:, c]) for c in range(22)])
            assert np.array_equal(pk[i], want)
            out = np.zeros(22 * PB, np.uint8)
            rc = lib.kmx_pack_row(sp[i].ctypes.data_as(ctypes.POINTER(ctypes.c_float)), X, Y, 22, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
            assert rc == 0 and np.array_equal(out, want)
    assert lib.kmx_pack_row(None, 19, 19, 22, None) != 0
