#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
run() { echo -n "$* : "; env "$@" timeout -k 10 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.read());print(round(o['value'],2), round(o['phases_ms']['factor'],4), o['check']['factor_ms']['min'])"; }
run MRBF_MEGA_CHAIN=16
run MRBF_MEGA_CHAIN=20
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_SLACK_CHAIN=7
run MRBF_MEGA_CHAIN=20 MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_CHAIN=24 MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_SLACK_CHAIN=5
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_SLACK_CHAIN=4
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_SROWS=3 MRBF_MEGA_SLACK_CHAIN=7
run MRBF_MEGA_CHAIN=20 MRBF_MEGA_SROWS=3
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_WIN=5
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_WIN=7 MRBF_MEGA_SLACK_CHAIN=7
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_LOOK=3
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_DEDICATED=48
run MRBF_MEGA_CHAIN=16 MRBF_MEGA_DEDICATED=96
