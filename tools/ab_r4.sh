#!/bin/bash
# alternating A/B of two environment settings of the persistent factorisation: tools/ab_r4.sh "<env A>" "<env B>" sizes reps rounds
A="$1"; B="$2"; SIZES=${3:-8192}; REPS=${4:-15}; ROUNDS=${5:-3}
for r in $(seq $ROUNDS); do
  echo -n "A [$A] : "; env $A timeout -k 10 300 python3 tools/potrf_time.py $SIZES $REPS 2>&1 | tail -1
  echo -n "B [$B] : "; env $B timeout -k 10 300 python3 tools/potrf_time.py $SIZES $REPS 2>&1 | tail -1
done
