#!/bin/bash
# hipGraph replay of a pass (KMX_GRAPHS=1) at SMALL batches (round 6): round 2 measured it at batch 256 only (nothing: a pass there is bound
# by its kernels). Device-resident passes at batch 1 / 8 / 24 / 32 / 64, A/B/A/B, then the self-play leg (8 x 8, 45 s) with and without.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/r06/graphs}
mkdir -p $OUT
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-pmc --no-profile "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/graphs.txt"; }
for rep in 1 2; do
  for n in 1 8 24 32 64; do
    b "batch $n direct launches" KMX_GRAPHS=0 -- --batch $n --steps 1500 --warmup 20
    b "batch $n graph replay   " KMX_GRAPHS=1 -- --batch $n --steps 1500 --warmup 20
  done
done
for g in 0 1 0 1; do
  KMX_GRAPHS=$g tools/selfplay_full_games.sh graphs_$g 8 8 8 8 45 > /dev/null 2>&1
  echo -n "self-play 8x8, KMX_GRAPHS=$g: " | tee -a $OUT/graphs.txt
  cut -c1-260 gpurun_out/selfplay_full_graphs_$g.txt | tail -1 | tee -a $OUT/graphs.txt
done
