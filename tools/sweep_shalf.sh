#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# half-height streamed jobs (MRBF_MEGA_SHALF = last so many block columns) against chain / reserve workgroup counts
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python3 tools/potrf_time.py $SIZES 9 2>&1 | tail -1; }
SIZES=2048,4096,4608,5120,6144
run MRBF_X=0
run MRBF_MEGA_SHALF=0 MRBF_MEGA_CHAIN=32
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=64
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=96
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=64 MRBF_MEGA_XCHAIN=2
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=48 MRBF_MEGA_XCHAIN=2
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=64 MRBF_MEGA_SLACK_CHAIN=5
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=64 MRBF_MEGA_SLACK_CHAIN=7
run MRBF_MEGA_SHALF=99 MRBF_MEGA_CHAIN=64 MRBF_MEGA_PSTREAM=3
