#!/bin/bash
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_r4 -o r4 -- python3 $ROOT/tools/round4_bench.py 64 10000 > $OUT/r4.txt 2>&1; echo "rc=$?"
cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof_r4/r4_results.db $OUT/round4_kernel_stats.csv 3 > /dev/null
head -14 $OUT/round4_kernel_stats.csv | cut -c1-150
find $OUT -name "*.db" -size +20M -delete
