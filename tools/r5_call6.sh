#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5f
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_pascoletti_serafini.py -x -q -m gpu -s > $OUT/pytest_ps.txt 2>&1; rc=$?; echo "pytest rc=$rc"
grep -E "PS ranking|PS step|passed|failed|Error|assert" $OUT/pytest_ps.txt | cut -c1-250 | tail -30
[ $rc -ne 0 ] && exit $rc
python3 tools/ps_bench2.py 64,128,256 > $OUT/ps_step.txt 2>&1; cat $OUT/ps_step.txt
