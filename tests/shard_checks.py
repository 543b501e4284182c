"""Checks of the .npz training shards `selfplay` writes (dataio/trainingwrite.h:166-349), shared by the CPU and the GPU tests of
BASELINE configs[4]: mixed board sizes in ONE data buffer, ownership and score targets, every row in exactly one shard."""
import glob
import os
import zipfile

import numpy as np

MEMBERS = ["binaryInputNCHWPacked", "globalInputNC", "policyTargetsNCMove", "globalTargetsNC", "scoreDistrN", "valueTargetsNCHW",
           "qValueTargetsNCMove"]  # trainingwrite.cpp:860-878, in this order


def shard_files(out_dir):
    return sorted(glob.glob(os.path.join(out_dir, "**", "tdata", "*.npz"), recursive=True))


def check_shards(files, L, board_sizes, rows_reported=None):
    """files: shards of one run with dataBoardLen = L; board_sizes: the square sizes the run may play (bSizes).
    Returns {size: rows} - how many training rows each board size contributed."""
    S = L * L
    per_size = {s: 0 for s in board_sizes}
    total = 0
    assert files, "selfplay wrote no training shard"
    for f in files:
        with zipfile.ZipFile(f) as z:
            assert z.testzip() is None
            assert [i.filename for i in z.infolist()] == MEMBERS
        with np.load(f) as z:
            n = z["globalInputNC"].shape[0]
            total += n
            want = {  # trainingwrite.h:180-349; inputs version 7, no metadata
                "binaryInputNCHWPacked": (np.uint8, (n, 22, (S + 7) // 8)),
                "globalInputNC": (np.float32, (n, 19)),
                "policyTargetsNCMove": (np.int16, (n, 2, S + 1)),
                "globalTargetsNC": (np.float32, (n, 80)),
                "scoreDistrN": (np.int8, (n, 2 * S + 120)),
                "valueTargetsNCHW": (np.int8, (n, 5, L, L)),
                "qValueTargetsNCMove": (np.int16, (n, 3, S + 1)),
            }
            for k, (dt, shape) in want.items():
                assert z[k].dtype == dt and z[k].shape == shape, (k, z[k].dtype, z[k].shape)
            planes = np.unpackbits(z["binaryInputNCHWPacked"], axis=2)[:, :, :S].reshape(n, 22, L, L)
            onboard = planes[:, 0]
            vt = z["valueTargetsNCHW"]
            pol = z["policyTargetsNCMove"]
            for b in range(n):
                ys, xs = np.nonzero(onboard[b])
                h, w = int(ys.max()) + 1, int(xs.max()) + 1
                assert h == w and h in per_size, (h, w)
                per_size[h] += 1
                # the board sits in the top-left corner of the LxL buffer, everything else is off the board
                assert onboard[b, :h, :w].all() and onboard[b].sum() == h * w
                assert not (planes[b, 1] & planes[b, 2]).any()          # own / opponent stones are disjoint
                assert not (planes[b, 1:] & (1 - onboard[b])).any()     # no feature off the board
                # ownership target (plane 0, in [-120, 120]) and the policy target live on the board only
                off = onboard[b] == 0
                assert not vt[b, 0][off].any() and np.abs(vt[b, 0].astype(np.int32)).max() <= 120
                assert not pol[b, 0, :S].reshape(L, L)[off].any()
            gt = z["globalTargetsNC"]
            assert np.all(np.isfinite(gt)) and np.all(np.abs(gt[:, 0:3].sum(axis=1) - 1.0) < 1e-5)  # win / loss / no-result
            assert np.all(pol[:, 0].sum(axis=1) > 0)
            assert np.all(z["scoreDistrN"].astype(np.int32).sum(axis=1) == 100)  # the score distribution sums to 100
    if rows_reported is not None:
        assert total == rows_reported, (total, rows_reported)
    return per_size
