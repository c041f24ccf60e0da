#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
set -o pipefail
OUT=gpurun_out/r5g
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_sampling.py -x -q -m gpu > $OUT/pytest_sampling.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_sampling.txt
[ $rc -ne 0 ] && exit $rc
for e in "MRBF_R4_EAGER=1 MRBF_R4_CUSTOM=1" "MRBF_R4_EAGER=0"; do echo "$e"; env $e python3 tools/round4_bench.py 64 10000 2>&1 | tail -2 | cut -c1-110; env $e python3 tools/round4_bench.py 128 6000 2>&1 | tail -1 | cut -c1-110;  env $e python3 tools/round4_bench.py 24 3000 2>&1 | tail -1 | cut -c1-110; done
