#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# streamed rows / chain workgroups of the persistent factorisation over problem sizes
for cfg in "MRBF_X=0" "MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=16" "MRBF_MEGA_SROWS=3 MRBF_MEGA_CHAIN=20" "MRBF_MEGA_SROWS=4 MRBF_MEGA_CHAIN=24" "MRBF_MEGA_SROWS=4 MRBF_MEGA_CHAIN=32" "MRBF_MEGA_SROWS=5 MRBF_MEGA_CHAIN=32"; do
  echo "== $cfg"; env $cfg timeout -k 10 200 python tools/factor_time.py 1024 2048 3072 4096 6144 8192 12288 2>&1 | grep "^n"
done
