#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# knobs of the persistent factorisation on a chain-bound size (C2: n = 2048, 16 block columns): factor phase of bench.py
run() { echo -n "$* : "; env "$@" timeout -k 10 100 python bench.py --config C2 --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.read());print(round(o['value'],1), round(o['phases_ms']['factor'],4))"; }
run MRBF_X=0
run MRBF_MEGA_SROWS=3
run MRBF_MEGA_SROWS=1
run MRBF_MEGA_CHAIN=16
run MRBF_MEGA_CHAIN=8
run MRBF_MEGA_SLACK_CHAIN=4
run MRBF_MEGA_SLACK_CHAIN=8
run MRBF_MEGA_WIN=6
run MRBF_MEGA_WIN=2
run MRBF_MEGA_DEDICATED=32
run MRBF_MEGA_DEDICATED=128
run MRBF_MEGA_LOOK=4
run MRBF_MEGA_GRID=256
run MRBF_MEGA_SLACK=2
run MRBF_MEGA_SLACK=5
