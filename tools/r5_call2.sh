#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5b
mkdir -p $OUT
rm -f $OUT/paneldma8.txt
for v in 1 8; do
echo "== MRBF_MEGA_PANELDMA=$v" >> $OUT/paneldma8.txt
MRBF_MEGA_PANELDMA=$v timeout -k 10 300 python3 tools/mega_check.py 1024,2048,4096 3 2 >> $OUT/paneldma8.txt 2>&1; echo "rc=$?"
done
cat $OUT/paneldma8.txt
