#!/bin/bash
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_ps256 -o ps -- python3 $ROOT/tools/ps_bench2.py 256 > $OUT/ps256.txt 2>&1; echo "rc=$?"
python3 $ROOT/tools/profile_summary.py stats $OUT/prof_ps256/ps_results.db $OUT/ps256_kernel_stats.csv 3 > /dev/null
head -12 $OUT/ps256_kernel_stats.csv | cut -c1-170
find $OUT -name "*.db" -size +20M -delete
