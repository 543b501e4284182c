#!/bin/bash
# Self-play on the 8 MI355X of one node, in ONE process - the reference's own multi-GPU mode: one NNEvaluator with a "server thread"
# (here: a leaf port = a persistent batcher) per GPU, chosen by <backend prefix>DeviceToUseThreadN (program/setup.cpp:174-220,
# nneval.cpp:399-407; cpp/configs/training/selfplay8mainb18.cfg:118-125 does the same with cudaDeviceToUseModel0Thread0..7).
# BASELINE configs[2] (8 parallel games per GPU) and the reference's production point (100 games per GPU: numGameThreads = 800, :65).
#
#   tools/selfplay_8gpu.sh <models-dir> <output-dir> [games per GPU = 8] [leaves per game = 8] [extra key=value ...]
#
# The config is the reference's production settings (tools/selfplay_cfg.py). Rows go to the device with the fewest rows in flight
# (integration/katamx_nneval.cpp pickPort). Host budget: ~185 us of host CPU per evaluated row (tests/test_host_capacity.py), i.e.
# ~7.4 cores per GPU at 40 k rows/s - 60 cores for the node; DESIGN.md section 6.
# The backend prefix of the build in this repository is the dummy backend's (the three cosmetic #elif of INTEGRATION.md section 2 are
# not applied to the unmodified reference): dummybackendDeviceToUseThreadN. A build with them uses katamxDeviceToUseThreadN.
# Not run on hardware in this repository's rounds (no 8-GPU node was available): the same command line runs on 8 FAKE devices in
# tests/test_schedule_dryrun.py::test_eight_devices_selfplay_in_one_process.
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
MODELS=$1; OUTDIR=$2; GAMES_PER_GPU=${3:-8}; LEAVES=${4:-8}
shift $(( $# < 4 ? $# : 4 ))
NGPU=${KMX_NUM_GPUS:-8}
PREFIX=${KMX_BACKEND_PREFIX:-dummybackend}
CFG=$(mktemp /tmp/selfplay_8gpu.XXXXXX.cfg)
KV=()
for ((i = 0; i < NGPU; i++)); do KV+=("${PREFIX}DeviceToUseThread$i=$i"); done
python3 "$REPO/tools/selfplay_cfg.py" "$CFG" numGameThreads=$((GAMES_PER_GPU * NGPU)) numSearchThreads=$LEAVES numNNServerThreadsPerModel=$NGPU \
  nnMaxBatchSize=256 "${KV[@]}" "$@" > /dev/null
export KATAMX_LEAVES_PER_THREAD=${KATAMX_LEAVES_PER_THREAD:-$LEAVES}
echo "config: $CFG ($NGPU devices, $((GAMES_PER_GPU * NGPU)) games, $LEAVES leaves per game, $KATAMX_LEAVES_PER_THREAD per OS thread)" >&2
exec ${KMX_LAUNCH_PREFIX:-} "$REPO/integration/_build/katago_hip" selfplay -config "$CFG" -models-dir "$MODELS" -output-dir "$OUTDIR" ${KMX_SELFPLAY_ARGS:-}
