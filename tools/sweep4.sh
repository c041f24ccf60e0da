#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# knobs of the persistent factorisation after the streamed fold in the S jobs (round 2), through bench.py's factor phase
run() { echo -n "$* : "; env "$@" timeout -k 10 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.read());print(round(o['value'],2), round(o['phases_ms']['factor'],4), o['check']['factor_ms']['min'])"; }
run MRBF_X=0
run MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_SLACK_CHAIN=10
run MRBF_MEGA_SLACK_CHAIN=12
run MRBF_MEGA_SROWS=3
run MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=16
run MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=16 MRBF_MEGA_SLACK_CHAIN=9
run MRBF_MEGA_CHAIN=16
run MRBF_MEGA_WIN=4
run MRBF_MEGA_WIN=4 MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_WIN=8 MRBF_MEGA_SLACK_CHAIN=10
run MRBF_MEGA_SLACK=4 MRBF_MEGA_SLACK_CHAIN=9
run MRBF_MEGA_SLACK=2 MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_DEDICATED=96 MRBF_MEGA_SLACK_CHAIN=9
run MRBF_MEGA_WBIAS=2 MRBF_MEGA_SLACK_CHAIN=9
run MRBF_MEGA_WBIAS=8 MRBF_MEGA_SLACK_CHAIN=9
