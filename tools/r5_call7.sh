#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5g
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_sampling.py -x -q -m gpu > $OUT/pytest_sampling.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_sampling.txt
[ $rc -ne 0 ] && exit $rc
python3 tools/round4_bench.py 64 10000 > $OUT/round4_d64.txt 2>&1; cat $OUT/round4_d64.txt
python3 tools/round4_bench.py 24 3000 > $OUT/round4_d24.txt 2>&1; cat $OUT/round4_d24.txt
