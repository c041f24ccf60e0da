#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# sweep of the persistent factorisation's scheduling knobs after the deferred bulk-update wait (round 2)
run() { echo "$@"; env "$@" timeout -k 10 100 python tools/mega_check.py 8192 3 4 2>&1 | grep "n="; }
run MRBF_X=0
for sc in 4 5 8 10; do run MRBF_MEGA_SLACK_CHAIN=$sc; done
for s in 2 4 5; do run MRBF_MEGA_SLACK=$s; done
for w in 4 8; do run MRBF_MEGA_WIN=$w; done
for ch in 9 15 18; do run MRBF_MEGA_CHAIN=$ch; done
for d in 32 96 128; do run MRBF_MEGA_DEDICATED=$d; done
for wb in 2 8; do run MRBF_MEGA_WBIAS=$wb; done
run MRBF_MEGA_SROWS=1
run MRBF_MEGA_SROWS=3
run MRBF_MEGA_LOOK=3
run MRBF_MEGA_QUIET=0
