#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
for cfg in "MRBF_MEGA_SROWS=6 MRBF_MEGA_CHAIN=40" "MRBF_MEGA_SROWS=7 MRBF_MEGA_CHAIN=48" "MRBF_MEGA_SROWS=5 MRBF_MEGA_CHAIN=32 MRBF_MEGA_DEDICATED=32" "MRBF_MEGA_SROWS=5 MRBF_MEGA_CHAIN=40 MRBF_MEGA_SLACK_CHAIN=8"; do
  echo "== $cfg"; env $cfg timeout -k 10 200 python tools/factor_time.py 1024 2048 4096 6144 2>&1 | grep "^n"
done
for cfg in "MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=24" "MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=20 MRBF_MEGA_SLACK_CHAIN=7" "MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=20 MRBF_MEGA_DEDICATED=48"; do
  echo "== $cfg"; env $cfg timeout -k 10 200 python tools/factor_time.py 8192 8192 2>&1 | grep "^n"
done
