#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5p
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_pascoletti_serafini.py -x -q -m gpu > $OUT/pytest_ps.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_ps.txt
[ $rc -ne 0 ] && exit $rc
python3 tools/ps_bench2.py 128,256 2>&1 | tail -6
MRBF_PS_MULTI=0 python3 tools/ps_bench2.py 128,256 2>&1 | tail -6
