#!/bin/bash
# batcher policy at the self-play operating points: linger before a partial batch goes, batches in flight
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c5; rm -rf $OUT; mkdir -p $OUT
run() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" bash tools/selfplay_full_games.sh $tag "$@" | sed "s/^/$tag [${envs[*]}]: /" | tee -a $OUT/batcher_policy.txt; }
run g100_l150 A=1 -- 100 1 1 100 40
run g100_l400 KMX_BATCH_LINGER_US=400 -- 100 1 1 100 40
run g100_l800 KMX_BATCH_LINGER_US=800 -- 100 1 1 100 40
run g100_l400_s3 KMX_BATCH_LINGER_US=400 -- 100 1 1 100 40 numNNServerThreadsPerModel=3
run g100x2_l400 KMX_BATCH_LINGER_US=400 -- 100 2 2 100 40
run g8x8_l150 A=1 -- 8 8 8 8 40
run g8x8_l400 KMX_BATCH_LINGER_US=400 -- 8 8 8 8 40
run g8x8_l800 KMX_BATCH_LINGER_US=800 -- 8 8 8 8 40
run g8x16_l400 KMX_BATCH_LINGER_US=400 -- 8 16 16 8 40
