#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# ablation of gram_w_kernel on the C3 workload under rocprofv3 (kernel time from the trace; an ablated kernel makes the fit fall
# back to the LU path, which does not matter here): MRBF_GW_DBG bits 1 no stores, 2 no radial function, 4 no W product, 8 no Gram
# product, 16 no staging; MRBF_GW_WGS workgroups (512 = two per CU).
ROOT=$(pwd)
export TMPDIR=/tmp
for v in "0 512" "1 512" "2 512" "3 512" "4 512" "8 512" "7 512" "16 512" "31 512" "0 256"; do
  set -- $v
  export MRBF_GW_DBG=$1 MRBF_GW_WGS=$2
  TAG=abl_$1_$2
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o t -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/$TAG.err)
  python3 tools/profile_summary.py stats gpurun_out/$TAG/t_results.db gpurun_out/$TAG.csv 4 > /dev/null
  python3 - <<PY
import csv, re
for r in csv.reader(open("gpurun_out/$TAG.csv")):
    if re.search(r"gram_w|proj_|chol_update_kernel<128", r[0]): print("dbg=$1 wgs=$2", r[0][:48].ljust(48), r[1], "avg %.1f us" % (float(r[3])/1e3), "min %.1f" % (float(r[5])/1e3))
PY
  rm -rf gpurun_out/$TAG gpurun_out/$TAG.csv
done
