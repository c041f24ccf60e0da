#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5b
mkdir -p $OUT
timeout -k 10 120 tools/gramlab/ovl 32 > $OUT/ovl.txt 2>&1; echo "ovl rc=$?"
cat $OUT/ovl.txt


