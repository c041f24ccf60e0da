#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5c
mkdir -p $OUT
timeout -k 10 300 tools/gramlab/gramlab 8192 8192 > $OUT/gramlab.txt 2>&1; echo "gramlab rc=$?"
cat $OUT/gramlab.txt
