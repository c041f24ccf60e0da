#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# schedule parameters of the persistent factorisation against each other (one process per setting; tools/potrf_time.py)
SIZES=${SIZES:-2048,8192}
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python3 tools/potrf_time.py $SIZES 7 2>&1 | tail -1; }
run MRBF_X=0
for v in "$@"; do run $v; done
