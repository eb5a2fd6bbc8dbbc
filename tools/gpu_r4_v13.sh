#!/bin/bash
# Round 4, visit 13: the consumer loop's recovery tests (team-kernel timeout record -> per-step kernels; f16 range -> bf16 pieces) and the
# ingest probe (where the ~2.6 ms per step go when the next batch is packed and copied beside the current one).
TAG=${1:-r4v13}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -q -x > $OUT/pytest_api.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/pytest_api.log
timeout 600 python tools/ingest_probe.py $OUT/ingest_probe.json > $OUT/ingest_probe.txt 2>&1
cat $OUT/ingest_probe.txt | tail -30
