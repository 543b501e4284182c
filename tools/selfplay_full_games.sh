#!/bin/bash
# games/hour as the reference defines it (command/selfplay.cpp:388-389: "Total games" x 3600 / "Total selfplay runtime (seconds)"), with
# FULL-LENGTH games played to the reference's end conditions at its production settings (tools/selfplay_cfg.py = selfplay8mainb18.cfg),
# b18c384nbt (random weights) on 19x19, through the product path (integration/_build/katago_hip: own evaluator + featuriser + fibers).
#   tools/selfplay_full_games.sh <tag> <game threads> <search threads per game> <leaves per OS thread> <max games total> <timeout s> [key=value ...]
# Writes gpurun_out/selfplay_full_<tag>.{log,txt}. A run that hits the timeout is interrupted with SIGINT (the reference then stops its games,
# writes its totals and exits cleanly): the .txt says so and reports rows/s only.
set -u
cd "$(dirname "$0")/.."
TAG=$1; GAMES=$2; SEARCH=$3; LEAVES=$4; MAXGAMES=$5; TMO=$6; shift 6
OUT=gpurun_out; mkdir -p $OUT
D=/tmp/sp_full_$TAG; rm -rf $D; mkdir -p $D/models
python3 - "$D" "$GAMES" "$SEARCH" "$@" <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import selfplay_cfg
from katago_amd import modelgen
d, games, search = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kv = dict(a.split("=", 1) for a in sys.argv[4:])
modelgen.write_model(d + "/models/b18c384nbt-s1-d1.bin.gz", "b18c384nbt", seed=7)
args = dict(numGameThreads=games, numSearchThreads=search, nnMaxBatchSize=256 if games * search >= 200 else 64, logGamesEvery=1 if games <= 16 else 10,
            switchNetsMidGame="false", nnCacheSizePowerOfTwo=21, nnMutexPoolSizePowerOfTwo=16, **selfplay_cfg.ONLY_19)
args.update(kv)
selfplay_cfg.write(d + "/main.cfg", **args)
PY
REPO=$PWD
( cd $D && KATAMX_LEAVES_PER_THREAD=$LEAVES timeout -s INT $TMO $REPO/integration/_build/katago_hip selfplay -config main.cfg -models-dir models -output-dir out -max-games-total $MAXGAMES > $REPO/$OUT/selfplay_full_$TAG.log 2>&1 )
python3 - "$OUT/selfplay_full_$TAG.log" "$TAG" "$GAMES" "$SEARCH" "$LEAVES" "$MAXGAMES" <<'PY' | tee $OUT/selfplay_full_$TAG.txt
import re, sys
t = open(sys.argv[1]).read()
tag, games_t, search, leaves, maxgames = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
g = lambda k: (re.findall(k + r": ([\d.]+)", t) or ["nan"])[-1]
secs, total = float(g(r"Total selfplay runtime \(seconds\)")), float(g("Total games"))
rows, moves, fin, batches, hits, drows = (float(g(k)) for k in ("Final NN rows", "Final moves played", "Final games finished", "Final NN batches", "Final NN cache hits", "Final data rows"))
interrupted = "Exited cleanly after signal" in t
head = "b18c384nbt 19x19 (random weights), katago_hip selfplay, production settings, %d game threads x %d search threads (%d per OS thread)" % (games_t, search, leaves)
if interrupted or fin < maxgames:
    print("%s: INTERRUPTED after %.0f s with %d of %d games finished, %d moves: %.0f NN rows/s (avg batch %.1f, %d cache hits)" % (head, secs, fin, maxgames, moves, rows / secs, rows / max(batches, 1), hits))
else:
    print("%s: %d games finished in %.1f s = %.1f games/hour (as the reference counts: 'Total games' %d x 3600 / runtime = %.1f); %.0f moves per game, "
          "%.0f NN rows per game (%.0f per move), %.0f NN rows/s, avg batch %.1f, %d cache hits, %d training rows"
          % (head, fin, secs, fin * 3600.0 / secs, total, total * 3600.0 / secs, moves / fin, rows / fin, rows / max(moves, 1), rows / secs, rows / max(batches, 1), hits, drows))
PY
