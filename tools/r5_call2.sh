#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
set -o pipefail
OUT=gpurun_out/r5b
mkdir -p $OUT
rm -f $OUT/paneldma_fix.txt
for v in 0 1 2; do
echo "== MRBF_MEGA_PANELDMA=$v (ring barrier = asm volatile s_barrier with memory clobber)" >> $OUT/paneldma_fix.txt
MRBF_MEGA_PANELDMA=$v timeout -k 10 300 python3 tools/mega_check.py 1024,2048,4096,8192 3 3 >> $OUT/paneldma_fix.txt 2>&1; echo "rc=$?"
done
cat $OUT/paneldma_fix.txt
