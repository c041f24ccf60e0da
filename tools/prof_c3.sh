#!/bin/bash
# rocprofv3 kernel trace of a short C3 bench run -> gpurun_out/<tag>.csv ; prints the kernels matching $2 (regex)
TAG=${1:-p}; PAT=${2:-backsolve|premul}; CFG=${3:-C3}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o t -- python3 $ROOT/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/$TAG.json 2> $ROOT/gpurun_out/$TAG.err
cd $ROOT
python3 tools/profile_summary.py stats gpurun_out/$TAG/t_results.db gpurun_out/$TAG.csv 9 > /dev/null
python3 - <<PY
import csv, re
for r in csv.reader(open("gpurun_out/$TAG.csv")):
    if re.search(r"$PAT", r[0]): print(r[0][:70].ljust(70), r[1], "avg %.1f us" % (float(r[3])/1e3), "min %.1f" % (float(r[5])/1e3), "max %.1f" % (float(r[6])/1e3))
PY
tail -c 420 gpurun_out/$TAG.json
