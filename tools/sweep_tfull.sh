#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# panel tiles as 128-row jobs (MRBF_MEGA_TFULL = block rows below the streamed ones that stay halves) against the default (all halves)
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python3 tools/potrf_time.py $SIZES 7 2>&1 | tail -1; }
SIZES=${SIZES:-4096,6400,8192,12288,16384}
run MRBF_X=0
for t in 0 1 2 4 8 16; do run MRBF_MEGA_TFULL=$t; done
run MRBF_X=0
