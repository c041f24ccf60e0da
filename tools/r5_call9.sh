#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5j
mkdir -p $OUT
bash tools/ab_r4.sh "MRBF_MEGA_PANELDMA=0" "MRBF_MEGA_PANELDMA=1" 1024,2048,4096,6144,8192,12288,16384 15 3 > $OUT/ab_paneldma.txt 2>&1
cat $OUT/ab_paneldma.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_schedules.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
