#!/bin/bash
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_r4b -o r4 -- python3 $ROOT/tools/round4_bench.py 128 6000 > $OUT/r4b.txt 2>&1; echo "rc=$?"
cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof_r4b/r4_results.db $OUT/round4_d128_kernel_stats.csv 3 > /dev/null
python3 - <<PY
import csv
rows=list(csv.reader(open("$OUT/round4_d128_kernel_stats.csv")))
for r in rows[1:14]: print(r[0][:90].ljust(90), r[1], "avg %.1f us" % (float(r[3])/1e3), "ms/call", r[8])
PY
find $OUT -name "*.db" -size +20M -delete
