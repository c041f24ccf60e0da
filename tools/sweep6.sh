#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
run() { echo -n "$* : "; env "$@" timeout -k 10 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.read());print(round(o['value'],2), round(o['phases_ms']['factor'],4), o['check']['factor_ms']['min'])"; }
run MRBF_X=0
run MRBF_MEGA_CHAIN=16
run MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=16
run MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=20
run MRBF_MEGA_SROWS=4 MRBF_MEGA_CHAIN=24
run MRBF_MEGA_SROWS=1 MRBF_MEGA_CHAIN=12
run MRBF_MEGA_SLACK=2
run MRBF_MEGA_SLACK=4
run MRBF_MEGA_WIN=8
run MRBF_MEGA_WIN=5
run MRBF_MEGA_DEDICATED=32
run MRBF_MEGA_DEDICATED=96
run MRBF_MEGA_HALF_COLS=2
run MRBF_MEGA_QUIET=0
