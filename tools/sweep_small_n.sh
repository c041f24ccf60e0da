#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# chain-bound sizes: streamed rows, stream depth, look-ahead against the defaults (halves on, 64 chain workgroups on four XCDs)
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python3 tools/potrf_time.py $SIZES 9 2>&1 | tail -1; }
SIZES=1024,2048,3072,4096
run MRBF_X=0
run MRBF_MEGA_SROWS=6
run MRBF_MEGA_SROWS=7
run MRBF_MEGA_SROWS=7 MRBF_MEGA_CHAIN=96 MRBF_MEGA_XCHAIN=4
run MRBF_MEGA_SROWS=4
run MRBF_MEGA_PSTREAM=3
run MRBF_MEGA_PSTREAM=1
run MRBF_MEGA_LOOK=2
run MRBF_MEGA_LOOK=6
run MRBF_MEGA_SLACK_CHAIN=4
run MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_WIN=2
run MRBF_MEGA_WIN=6
run MRBF_X=0
