#!/bin/bash
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_ps -o ps -- python3 $ROOT/tools/ps_bench2.py 256 > $OUT/ps256.txt 2>&1; echo "ps rc=$?"
cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof_ps/ps_results.db $OUT/ps256_kernel_stats.csv 3 > /dev/null
head -16 $OUT/ps256_kernel_stats.csv | cut -c1-160
cat $OUT/ps256.txt | tail -4
find $OUT -name "*.db" -size +20M -delete
