#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MRBF_R4_EAGER=-1
for cfg in "64 10000" "128 6000"; do
tag=$(echo $cfg | tr ' ' '_')
rocprofv3 --kernel-trace --stats -d $OUT/prof_e$tag -o r4 -- python3 $ROOT/tools/round4_bench.py $cfg > $OUT/r4e$tag.txt 2>&1; echo "rc=$?"
python3 $ROOT/tools/profile_summary.py stats $OUT/prof_e$tag/r4_results.db $OUT/round4_e${tag}_kernel_stats.csv 3 > /dev/null
python3 - <<PY
import csv
rows=list(csv.reader(open("$OUT/round4_e${tag}_kernel_stats.csv")))
print("$cfg")
for r in rows[1:9]: print("  ", r[0][:80].ljust(80), r[1], "avg %.1f us" % (float(r[3])/1e3), "ms/call", r[8])
PY
done
find $OUT -name "*.db" -size +20M -delete
