"""-m gpu: the reference's `selfplay` command — the one its games/hour metric is defined on (command/selfplay.cpp:388-389)
— on the HIP backend (integration/_build/katago_hip), writing .npz shards through integration/zipfile_zlib.cpp. The CPU twin of
this test (tests/test_npz_writer.py) runs the same command on the oracle; here the rows are evaluated on the MI355X on a
9x9 buffer with 7x7 boards masked inside it, in greedy batches of up to 8 rows from 8 game threads."""
import glob
import os
import shutil
import subprocess
import zipfile

import numpy as np
import pytest

import shard_checks
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
    assert "katamx (HIP/gfx950)" in log and "CPU oracle" not in log
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


def test_selfplay_rate_b18_19x19_on_hip(tmp_path):
    """The second half of BASELINE's metric: games/hour = "Total games" * 3600 / "Total selfplay runtime (seconds)"
    (command/selfplay.cpp:388-389) for b18c384nbt (random weights) on 19x19 through the reference's `selfplay` on the HIP
    backend - BASELINE configs[2]'s 8 parallel games per GPU, and 128 game threads (batches the device can use). Games are
    capped at 60 moves and 32 / 16 visits so that the test takes seconds: what is recorded (gpurun_out/, copied to profiles/)
    is the rate AT THAT SETTING together with the NN rows per second it implies - not a full-length self-play figure."""
    from katago_amd import modelgen

    b = ref_binary("katago_hip_refeval")  # the reference's own evaluator: its server threads are what this test varies
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "models"))
    modelgen.write_model(os.path.join(d, "models", "b18c384nbt-s1-d1.bin.gz"), "b18c384nbt", seed=7)
    lines = []
    for threads, servers in ((8, 1), (8, 4), (128, 1)):
        out = os.path.join(d, "out%d_%d" % (threads, servers))
        over = ("numGameThreads=%d,nnMaxBatchSize=%d,dataBoardLen=19,bSizes=19,bSizeRelProbs=1,maxMovesPerGame=60,maxVisits=32,"
                "cheapSearchVisits=16,reducedVisitsMin=16,maxRowsPerTrainFile=20000,maxDataQueueSize=2000,nnCacheSizePowerOfTwo=18,"
                "nnMutexPoolSizePowerOfTwo=14,logGamesEvery=1000,numNNServerThreadsPerModel=%d" % (threads, max(threads, 8), servers))
        p = subprocess.run([b, "selfplay", "-config", CFG, "-models-dir", os.path.join(d, "models"), "-output-dir", out,
                            "-max-games-total", str(threads), "-override-config", over], capture_output=True, text=True, timeout=600, cwd=d)
        log = p.stdout + p.stderr
        assert p.returncode == 0 and "All cleaned up, quitting" in log, log[-3000:]
        games = int(log.split("Total games: ")[1].split()[0])
        secs = float(log.split("Total selfplay runtime (seconds): ")[1].split()[0])
        nn_rows = int(log.split("Final NN rows: ")[1].split()[0])
        assert games >= threads and secs > 0 and nn_rows > 0
        lines.append("b18c384nbt 19x19 selfplay, %3d game threads, %d NN server thread(s), <=60 moves, 32/16 visits: %d games in %.1f s = %.0f games/hour; %d NN rows = %.0f rows/s"
                     % (threads, servers, games, secs, games * 3600.0 / secs, nn_rows, nn_rows / secs))
    print("\n".join(lines))
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "selfplay_rate_b18.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")


def test_selfplay_rate_with_leaves_in_flight_per_game(tmp_path):
    """BASELINE configs[2] (8 parallel games per GPU) with this repo's NNEvaluator (no server threads: a game thread hands its
    rows to the leaf batcher itself) and then with 8 leaves in flight PER GAME: numSearchThreads = 8, the 8 search threads of a
    game running as fibers on the game's own OS thread (integration/katamx_fibers.cpp) - still 8 OS threads and 8 games, but 64
    rows per device batch instead of 8. Round 2 measured 766 - 1 176 NN rows/s at this operating point through the reference's
    evaluator (2.4 ms per pass whatever the batch is below 32 rows); VERDICT asked for >= 2.5 k."""
    from katago_amd import modelgen

    b = ref_binary("katago_hip")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "models"))
    modelgen.write_model(os.path.join(d, "models", "b18c384nbt-s1-d1.bin.gz"), "b18c384nbt", seed=7)
    lines = []
    rates = {}
    for name, search_threads, leaves, visits, cheap in (("1 leaf per game", 1, 1, 32, 16), ("8 leaves per game (fibers)", 8, 8, 32, 16),
                                                       ("8 leaves per game (fibers), visits of selfplay8mainb18.cfg", 8, 8, 2000, 350)):
        out = os.path.join(d, "out%d_%d" % (leaves, visits))
        moves = 60 if visits <= 32 else 6  # the real visit counts make a move cost ~760 rows: a few moves per game keep the test short
        over = ("numGameThreads=8,numSearchThreads=%d,nnMaxBatchSize=64,dataBoardLen=19,bSizes=19,bSizeRelProbs=1,maxMovesPerGame=%d,maxVisits=%d,"
                "cheapSearchVisits=%d,reducedVisitsMin=%d,maxRowsPerTrainFile=20000,maxDataQueueSize=2000,nnCacheSizePowerOfTwo=18,"
                "nnMutexPoolSizePowerOfTwo=14,logGamesEvery=1000,numNNServerThreadsPerModel=2" % (search_threads, moves, visits, cheap, cheap))
        env = dict(os.environ, KATAMX_LEAVES_PER_THREAD=str(leaves), KATAMX_FIBER_STATS="1")
        p = subprocess.run([b, "selfplay", "-config", CFG, "-models-dir", os.path.join(d, "models"), "-output-dir", out,
                            "-max-games-total", "8", "-override-config", over], capture_output=True, text=True, timeout=600, cwd=d, env=env)
        log = p.stdout + p.stderr
        assert p.returncode == 0 and "All cleaned up, quitting" in log, log[-3000:]
        games = int(log.split("Total games: ")[1].split()[0])
        secs = float(log.split("Total selfplay runtime (seconds): ")[1].split()[0])
        nn_rows = int(log.split("Final NN rows: ")[1].split()[0])
        assert games >= 8 and secs > 0 and nn_rows > 0
        rates[name] = nn_rows / secs
        lines.append("b18c384nbt 19x19 selfplay, 8 game threads, %s, <=%d moves, %d/%d visits: %d games in %.1f s = %.0f games/hour; %d NN rows = %.0f rows/s"
                     % (name, moves, visits, cheap, games, secs, games * 3600.0 / secs, nn_rows, nn_rows / secs))
    print("\n".join(lines))
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "selfplay_rate_b18_own_evaluator.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    # absolute floors beside the ratio below (a relaxed ratio must not hide an absolute regression): round 4's last full run
    # measured 1 207 / 3 639 / 8 883 rows/s on a slow box; the floors are those minus the spread between boxes (~25 %)
    assert rates["8 leaves per game (fibers), visits of selfplay8mainb18.cfg"] >= 6000.0, lines
    assert rates["8 leaves per game (fibers)"] >= 2600.0 and rates["1 leaf per game"] >= 850.0, lines
    # (2.9-3.9 x in rounds 3-4; 1.94 x once small passes got faster at the end of round 4 - 2.5 k -> 4.9 k rows/s: a pass over 8 rows
    # fell from 2.07 to 1.65 ms, a pass over ~30 rows did not - both rates rose)
    assert rates["8 leaves per game (fibers)"] >= 1.5 * rates["1 leaf per game"], lines


def test_mixed_board_sizes_b18_own_evaluator_writes_valid_shards(tmp_path):
    """BASELINE configs[4] through the PRODUCT path: `katago_hip selfplay` (this repo's NNEvaluator + featuriser + fibers over the leaf
    batcher) on b18c384nbt with the reference's production self-play settings (tools/selfplay_cfg.py = selfplay8mainb18.cfg), board
    sizes 9 / 13 / 19 mixed in ONE 19x19 buffer (selfplay8mainb18.cfg:76-77 mixes 13 sizes the same way), 8 games x 8 leaves in
    flight, ownership + score targets, .npz shards. Every row the evaluator submits is also featurised by the reference's
    NNInputs::fillRowV7 and compared (KATAMX_FEATURES=check aborts on the first difference). Visits are cut to 48 / 24 and games
    to 50 moves so that the test takes a minute; the rates of full-length games are bench.py's and profiles/' business."""
    import sys

    sys.path.insert(0, os.path.join(REPO, "tools"))
    import selfplay_cfg
    from katago_amd import modelgen

    b = ref_binary("katago_hip")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "models"))
    modelgen.write_model(os.path.join(d, "models", "b18c384nbt-s1-d1.bin.gz"), "b18c384nbt", seed=7)
    cfg = selfplay_cfg.write(os.path.join(d, "mixed.cfg"), numGameThreads=8, numSearchThreads=8, nnMaxBatchSize=64, maxVisits=48,
                             cheapSearchVisits=24, reducedVisitsMin=24, estimateLeadVisits=6, maxMovesPerGame=50, logGamesEvery=1000,
                             nnCacheSizePowerOfTwo=18, nnMutexPoolSizePowerOfTwo=14, maxRowsPerTrainFile=400, firstFileRandMinProp=1.0,
                             switchNetsMidGame="false", handicapAsymmetricPlayoutProb=0.0, normalAsymmetricPlayoutProb=0.0,  # (a visit count divided by up to 8 must stay >= 5)
                             **selfplay_cfg.MIXED_9_13_19)
    env = dict(os.environ, KATAMX_LEAVES_PER_THREAD="8", KATAMX_FEATURES="check")
    p = subprocess.run([b, "selfplay", "-config", cfg, "-models-dir", os.path.join(d, "models"), "-output-dir", os.path.join(d, "out"),
                        "-max-games-total", "30"], capture_output=True, text=True, timeout=900, cwd=d, env=env)
    log = p.stdout + p.stderr
    assert p.returncode == 0 and "All cleaned up, quitting" in log, log[-3000:]
    assert "katamx (HIP/gfx950)" in log and "CPU oracle" not in log
    rows = int(log.split("Final data rows: ")[1].split()[0])
    nn_rows = int(log.split("Final NN rows: ")[1].split()[0])
    batches = int(log.split("Final NN batches: ")[1].split()[0])
    games = int(log.split("Final games finished: ")[1].split()[0])
    secs = float(log.split("Total selfplay runtime (seconds): ")[1].split()[0])
    assert games >= 30 and rows > 0 and nn_rows > 20 * games
    per_size = shard_checks.check_shards(shard_checks.shard_files(os.path.join(d, "out")), 19, (9, 13, 19), rows)
    line = ("configs[4]: katago_hip selfplay b18c384nbt, bSizes 9,13,19 in a 19x19 buffer, 8 games x 8 leaves, 48/24 visits, <=50 moves, "
            "KATAMX_FEATURES=check: %d games, %d training rows (by board size %s), %d NN rows in %d batches (avg %.1f), %.1f s = %.0f NN rows/s"
            % (games, rows, per_size, nn_rows, batches, nn_rows / max(batches, 1), secs, nn_rows / secs))
    print(line)
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "selfplay_mixed_sizes_b18.txt"), "w") as f:
            f.write(line + "\n")
    assert all(v > 0 for v in per_size.values()), per_size  # all three board sizes produced training rows
