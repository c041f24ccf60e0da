#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5l
mkdir -p $OUT
for v in 0 1; do
echo "== MRBF_SMALL_DIAG6=$v"
MRBF_SMALL_DIAG6=$v MRBF_SMALL_STAMPS=1 timeout -k 5 200 python3 tools/small_stamps.py 2>&1 | grep -E "n=257 d=128|n=512|n=100" | tail -4 | cut -c1-230
MRBF_SMALL_DIAG6=$v timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
MRBF_SMALL_DIAG6=$v python3 bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4', round(d['value']), d['one_of_eight_gpus'])"
done
