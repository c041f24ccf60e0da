#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# n = 8192 (fit-shaped): streamed tiles as halves in the last block columns, with enough reserve chain workgroups on their own CUs
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python3 tools/fit_factor_time.py ${SIZES:-8192} 2>&1 | tail -1; }
run MRBF_X=0
for sh in 12 16 20 24; do for rs in 28 44; do for xc in 2 4; do run MRBF_MEGA_SHALF=$sh MRBF_MEGA_RESERVE=$rs MRBF_MEGA_XCHAIN=$xc; done; done; done
run MRBF_X=0
run MRBF_MEGA_SHALF=20 MRBF_MEGA_RESERVE=44 MRBF_MEGA_XCHAIN=4 MRBF_MEGA_TAIL=20
run MRBF_MEGA_SHALF=24 MRBF_MEGA_RESERVE=44 MRBF_MEGA_XCHAIN=4 MRBF_MEGA_TAIL=24 MRBF_MEGA_TAILHALF=28
