#!/bin/bash
# sampling tests, then round-4 timings (three repetitions per shape)
set -o pipefail
OUT=gpurun_out/r5g
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_sampling.py -x -q -m gpu > $OUT/pytest_sampling.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_sampling.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do python3 tools/round4_bench.py 64 10000 2>&1 | tail -1 | cut -c1-110; python3 tools/round4_bench.py 128 6000 2>&1 | tail -1 | cut -c1-110; done
python3 tools/round4_bench.py 24 3000 2>&1 | tail -1 | cut -c1-110
bash tools/r5_call8.sh > $OUT/trace.txt 2>&1; grep -v "Cijk\|eval_fused\|potrf" $OUT/trace.txt | cut -c1-150
