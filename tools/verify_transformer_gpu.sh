#!/bin/bash
# First GPU call of the next round: run the transformer device path, which was written
# against the oracle without hardware, and keep everything needed to debug it in gpurun_out/transformer/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/verify_transformer_gpu.sh'
# Order: unit kernels first (rmsnorm, attention, swiglu against numpy), then whole nets (PyTorch goldens, the
# reference's trained transformer test nets against the oracle), then the convolutional regression suite.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/transformer
mkdir -p "$OUT"
export KMX_EXPERIMENTAL_TRANSFORMER=1
python -m pytest tests/test_gpu_transformer.py -m gpu -q -k "kernel" -p no:cacheprovider 2>&1 | tail -60 > "$OUT/unit.log"
python -m pytest tests/test_gpu_transformer.py -m gpu -q -s -k "not kernel" -p no:cacheprovider 2>&1 | tail -120 > "$OUT/nets.log"
unset KMX_EXPERIMENTAL_TRANSFORMER
python -m pytest tests/test_gpu_model.py tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > "$OUT/conv_regression.log"
tail -5 "$OUT/unit.log" "$OUT/nets.log" "$OUT/conv_regression.log"
