#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# Parameter sweeps of the persistent factorisation (environment knobs of chol_mega.hip / context.hip), e.g.
#   MRBF_MEGA_SLACK, MRBF_MEGA_SLACK_CHAIN, MRBF_MEGA_WIN, MRBF_MEGA_WBIAS, MRBF_MEGA_DEDICATED, MRBF_MEGA_CHAIN, MRBF_MEGA_LOOK,
#   MRBF_MEGA_FIRST_WINDOW, MRBF_MEGA_SROWS; MRBF_MEGA_DEBUG=1 prints which dependency a bounded spin gave up on.
for wb in 0 2 4 8; do
    echo "wbias=$wb"
    MRBF_MEGA_WBIAS=$wb timeout -k 10 200 python tools/mega_check.py 8192,16384 3 3 2>&1 | grep "n="
done
