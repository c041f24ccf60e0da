#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# combinations of the persistent factorisation's knobs measured through bench.py's factor phase (round 2)
run() { echo -n "$* : "; env "$@" timeout -k 10 100 python bench.py --steps 80 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.read());print(round(o['value'],2), round(o['phases_ms']['factor'],4), o['check']['factor_ms']['min'])"; }
run MRBF_X=0
run MRBF_MEGA_SLACK=5 MRBF_MEGA_DEDICATED=96
run MRBF_MEGA_SLACK=5 MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_SLACK=4 MRBF_MEGA_SLACK_CHAIN=5
run MRBF_MEGA_SLACK=5 MRBF_MEGA_WIN=8
run MRBF_MEGA_SLACK=6 MRBF_MEGA_SLACK_CHAIN=8 MRBF_MEGA_WIN=8
run MRBF_MEGA_SLACK=5 MRBF_MEGA_LOOK=3 MRBF_MEGA_DEDICATED=128
run MRBF_MEGA_SLACK=5 MRBF_MEGA_CHAIN=15
run MRBF_MEGA_FIRST_WINDOW=2
run MRBF_MEGA_SLACK=5 MRBF_MEGA_HALF_COLS=2
