#!/bin/bash
# Round 6, last probe: what an MFMA-only kernel sustains on operands DISTRIBUTED like the bench's own (kind 3 of kmx_bench_mfma_sustained) beside
# uniform noise (kind 2) - and, since the library was rebuilt for it, smoke() and the whole-net parity file on the rebuilt library.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/last_r06; mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 200 python tools/mfma_power_probe.py 2.0 2>&1 | grep "mfma power" | tee $OUT/mfma_power_netlike.txt
timeout 200 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_command.py::test_sustained_mfma_rate_depends_on_the_operands -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_model_rebuilt.txt
