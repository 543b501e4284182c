"""-m gpu: the reference's `selfplay` command — the one its games/hour metric is defined on (command/selfplay.cpp:388-389)
— on the HIP backend (oracle/_ref/katago_hip), writing .npz shards through integration/zipfile_zlib.cpp. The CPU twin of
this test (tests/test_npz_writer.py) runs the same command on the oracle; here the rows are evaluated on the MI355X on a
9x9 buffer with 7x7 boards masked inside it, in greedy batches of up to 8 rows from 8 game threads."""
import glob
import os
import shutil
import subprocess
import zipfile

import numpy as np
import pytest

from conftest import REPO, ref_binary

pytestmark = pytest.mark.gpu
G170 = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")
CFG = os.path.join(REPO, "tests", "configs", "selfplay_tiny.cfg")


def test_selfplay_writes_shards_on_hip(tmp_path):
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    b = ref_binary("katago_hip")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "models"))
    shutil.copy(G170, os.path.join(d, "models"))
    p = subprocess.run([b, "selfplay", "-config", CFG, "-models-dir", os.path.join(d, "models"), "-output-dir", os.path.join(d, "out"),
                        "-max-games-total", "8", "-override-config", "numGameThreads=8"],
                       capture_output=True, text=True, timeout=900, cwd=d)
    log = p.stdout + p.stderr
    assert p.returncode == 0 and "All cleaned up, quitting" in log, log[-3000:]
    assert "katamx (HIP/gfx950) backend" in log
    rows = int(log.split("Final data rows: ")[1].split()[0])
    nn_rows = int(log.split("Final NN rows: ")[1].split()[0])
    games = int(log.split("Final games finished: ")[1].split()[0])
    assert games >= 8 and nn_rows > 100 * games and rows > 0
    files = sorted(glob.glob(os.path.join(d, "out", "**", "tdata", "*.npz"), recursive=True))
    assert files
    total = 0
    for f in files:
        with zipfile.ZipFile(f) as z:
            assert z.testzip() is None
        with np.load(f) as z:
            n = z["globalInputNC"].shape[0]
            total += n
            assert z["binaryInputNCHWPacked"].shape == (n, 22, 11) and z["policyTargetsNCMove"].shape == (n, 2, 82)
            gt = z["globalTargetsNC"]
            assert np.all(np.isfinite(gt)) and np.all(np.abs(gt[:, 0:3].sum(axis=1) - 1.0) < 1e-5)
    assert total == rows
