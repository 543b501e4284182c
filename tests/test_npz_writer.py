"""the on-disk training-data format. KataGo's `selfplay` writes .npz shards through `ZipFile`
(dataio/numpywrite.h:45-61), which only exists over libzip; integration/zipfile_zlib.cpp supplies it over zlib. Here the
reference's own `selfplay` command (reference host code, bound to the CPU oracle: oracle/_ref/katago_oracle) plays a few
tiny games and the shards it writes are read back with Python's zipfile / numpy — the readers the reference's training
code uses (python/katago/train/data_processing_pytorch.py:43-97) — and checked against the layouts
dataio/trainingwrite.h:166-349 documents. The packed input planes of every row are also re-derived with kmx_pack_row:
the product's packed-row entry (kmx_eval_packed) takes exactly the bytes the reference stores on disk."""
import glob
import os
import shutil
import subprocess
import sys
import zipfile

import numpy as np
import pytest

from conftest import REPO, REF, REF_BIN_DIR

BIN = os.path.join(REF_BIN_DIR, "katago_oracle")
MODEL = os.path.join(REF_BIN_DIR, "models", "g170-b6c96-s175395328-d26788732.bin.gz")
CFG = os.path.join(REPO, "tests", "configs", "selfplay_tiny.cfg")
MEMBERS = ["binaryInputNCHWPacked", "globalInputNC", "policyTargetsNCMove", "globalTargetsNC", "scoreDistrN", "valueTargetsNCHW",
           "qValueTargetsNCMove"]  # trainingwrite.cpp:860-878, in this order


@pytest.fixture(scope="module")
def shards(tmp_path_factory):
    if not (os.path.exists(BIN) and os.path.exists(MODEL)):
        pytest.skip("oracle/_ref/katago_oracle not built (make -C oracle ref needs the reference checkout)")
    d = str(tmp_path_factory.mktemp("selfplay"))
    os.makedirs(os.path.join(d, "models"))
    shutil.copy(MODEL, os.path.join(d, "models"))
    p = subprocess.run([BIN, "selfplay", "-config", CFG, "-models-dir", os.path.join(d, "models"), "-output-dir", os.path.join(d, "out"),
                        "-max-games-total", "3"], capture_output=True, text=True, timeout=900, cwd=d)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    log = p.stdout + p.stderr
    assert "All cleaned up, quitting" in log
    rows = int(log.split("Final data rows: ")[1].split()[0])
    files = sorted(glob.glob(os.path.join(d, "out", "**", "tdata", "*.npz"), recursive=True))
    assert files, "selfplay wrote no training shard"
    assert not glob.glob(os.path.join(d, "out", "**", "*.tmp*"), recursive=True)  # shards are renamed into place after close()
    return files, rows


def test_shards_are_valid_deflated_zip_archives(shards):
    files, _ = shards
    for f in files:
        with zipfile.ZipFile(f) as z:
            assert z.testzip() is None  # every member inflates and matches its CRC-32
            assert [i.filename for i in z.infolist()] == MEMBERS
            for i in z.infolist():
                assert i.compress_type == zipfile.ZIP_DEFLATED and i.compress_size < i.file_size
                raw = z.read(i.filename)
                assert raw[:6] == b"\x93NUMPY" and (len(raw) - 256) >= 0  # NumpyBuffer: 256-byte header (numpywrite.h:26)


def test_shard_arrays_have_the_documented_layout(shards):
    files, rows = shards
    total = 0
    L, S = 9, 81
    for f in files:
        with np.load(f) as z:
            assert sorted(z.files) == sorted(MEMBERS)
            n = z["globalInputNC"].shape[0]
            total += n
            want = {  # trainingwrite.h:180-349; dataBoardLen 9, inputs version 7, no metadata
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
            for b in range(n):
                ys, xs = np.nonzero(onboard[b])
                h, w = ys.max() + 1, xs.max() + 1
                assert (h, w) in ((9, 9), (7, 7)) and onboard[b, :h, :w].all() and onboard[b].sum() == h * w  # bSizes = 9,7
                assert not (planes[b, 1] & planes[b, 2]).any()  # own / opponent stones are disjoint
                assert not (planes[b, 1:] & (1 - onboard[b])).any()  # nothing off-board
            gt = z["globalTargetsNC"]
            assert np.all(np.abs(gt[:, 0:3].sum(axis=1) - 1.0) < 1e-5)  # win/loss/noresult targets are a distribution
            assert np.all(z["policyTargetsNCMove"][:, 0].sum(axis=1) > 0)
            assert np.all(z["scoreDistrN"].astype(np.int32).sum(axis=1) == 100)  # trainingwrite.h: score distribution sums to 100
    assert total == rows  # every row selfplay reported is in exactly one shard


def test_on_disk_planes_are_the_packed_rows_the_backend_accepts(shards):
    import ctypes

    from katago_amd import capi

    lib = capi.load_library()
    files, _ = shards
    with np.load(files[0]) as z:
        packed = z["binaryInputNCHWPacked"]
    n = min(8, packed.shape[0])
    planes = np.unpackbits(packed[:n], axis=2)[:, :, :81]  # [n, 22, 81] NCHW
    for b in range(n):
        nhwc = np.ascontiguousarray(planes[b].T.astype(np.float32))  # rowSpatialBuf layout: [81][22]
        out = np.zeros(22 * 11, np.uint8)
        rc = lib.kmx_pack_row(nhwc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 9, 9, 22, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        assert rc == 0 and np.array_equal(out.reshape(22, 11), packed[b])


def test_reference_training_loader_reads_the_shards(shards):
    if not os.path.isdir(os.path.join(REF, "python", "katago")):
        pytest.skip("reference checkout not present")
    files, rows = shards
    code = r"""
import sys
sys.path.insert(0, sys.argv[1])
from katago.train import data_processing_pytorch as dp, modelconfigs
cfg = modelconfigs.config_of_name["b6c96"]
n = 0
for batch in dp.read_npz_training_data(sys.argv[2:], 4, 1, 0, 9, "cpu", False, False, cfg):
    assert batch["binaryInputNCHW"].shape == (4, 22, 9, 9) and batch["globalInputNC"].shape == (4, 19)
    assert batch["policyTargetsNCMove"].shape == (4, 2, 82) and batch["globalTargetsNC"].shape == (4, 80)
    n += 4
print("ROWS", n)
"""
    p = subprocess.run([sys.executable, "-c", code, os.path.join(REF, "python")] + files, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    got = int(p.stdout.split("ROWS")[1])
    with_whole_batches = 0
    for f in files:
        with np.load(f) as z:
            with_whole_batches += (z["globalInputNC"].shape[0] // 4) * 4
    assert got == with_whole_batches and got > rows // 2


def test_zipfile_on_its_own(tmp_path):
    """integration/zipfile_selftest.cpp: empty member, a 3 MB member, a replaced member (ZIP_FL_OVERWRITE semantics),
    a partial-batch NumpyBuffer, an archive that is never closed (discarded, as zip_discard does) and an unwritable path
    (StringError)."""
    exe = os.path.join(REF_BIN_DIR, "zipfile_selftest")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/zipfile_selftest not built")
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "closed ok.zip" in p.stdout and "threw: Could not open zip file" in p.stdout
    assert sorted(os.listdir(tmp_path)) == ["ok.zip"]  # the abandoned archive is gone
    with zipfile.ZipFile(os.path.join(tmp_path, "ok.zip")) as z:
        assert z.testzip() is None
        assert [i.filename for i in z.infolist()] == ["empty", "big", "replaced", "partial"]
        assert z.read("empty") == b""
        assert z.read("replaced") == bytes([9]) * 2000
        s, big = 12345, bytearray(3 * 1000 * 1000 + 17)
        state = np.empty(len(big), np.uint32)
        for i in range(len(big)):  # the driver's LCG
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            state[i] = s
        mask = np.where((np.arange(len(big)) // 4096) % 2 == 1, 0x0F, 0xFF).astype(np.uint32)
        assert z.read("big") == (((state >> 24) & mask).astype(np.uint8)).tobytes()
    with np.load(os.path.join(tmp_path, "ok.zip")) as z:
        a = z["partial"]
        assert a.dtype == np.float32 and a.shape == (3, 3, 2) and np.array_equal(a.reshape(-1), 0.5 * np.arange(18, dtype=np.float32))


def test_mixed_board_sizes_in_one_buffer_cpu_twin(tmp_path):
    """The CPU twin of tests/test_gpu_selfplay.py::test_mixed_board_sizes_b18_own_evaluator_writes_valid_shards (BASELINE configs[4]):
    the reference's production self-play settings (tools/selfplay_cfg.py) with 9 / 13 / 19 boards mixed in one 19x19 buffer, through
    this repo's NNEvaluator and featuriser (oracle/_ref/katago_oraclex: the same host code over the CPU oracle instead of the
    MI355X), every submitted row compared with NNInputs::fillRowV7 (KATAMX_FEATURES=check); the small g170 net and a handful of
    visits keep it to seconds. The shards are checked for all three sizes, ownership / policy targets confined to the board."""
    import shard_checks

    sys.path.insert(0, os.path.join(REPO, "tools"))
    import selfplay_cfg

    bx = os.path.join(REF_BIN_DIR, "katago_oraclex")
    if not (os.path.exists(bx) and os.path.exists(MODEL)):
        pytest.skip("oracle/_ref/katago_oraclex not built (make -C oracle ref needs the reference checkout)")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "models"))
    shutil.copy(MODEL, os.path.join(d, "models"))
    cfg = selfplay_cfg.write(os.path.join(d, "mixed.cfg"), numGameThreads=4, numSearchThreads=2, nnMaxBatchSize=8, maxVisits=6,
                             cheapSearchVisits=3, reducedVisitsMin=3, estimateLeadVisits=2, maxMovesPerGame=24, logGamesEvery=1000,
                             nnCacheSizePowerOfTwo=14, nnMutexPoolSizePowerOfTwo=10, maxRowsPerTrainFile=60, firstFileRandMinProp=1.0,
                             switchNetsMidGame="false", handicapAsymmetricPlayoutProb=0.0, normalAsymmetricPlayoutProb=0.0,  # (a visit count divided by up to 8 must stay >= 5)
                             **selfplay_cfg.MIXED_9_13_19)
    env = dict(os.environ, KATAMX_FEATURES="check", KATAMX_LEAVES_PER_THREAD="2")
    p = subprocess.run([bx, "selfplay", "-config", cfg, "-models-dir", os.path.join(d, "models"), "-output-dir", os.path.join(d, "out"),
                        "-max-games-total", "24"], capture_output=True, text=True, timeout=900, cwd=d, env=env)
    log = p.stdout + p.stderr
    assert p.returncode == 0 and "All cleaned up, quitting" in log, log[-3000:]
    rows = int(log.split("Final data rows: ")[1].split()[0])
    per_size = shard_checks.check_shards(shard_checks.shard_files(os.path.join(d, "out")), 19, (9, 13, 19), rows)
    print("training rows by board size:", per_size)
    assert all(v > 0 for v in per_size.values()), per_size
