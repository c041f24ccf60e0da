#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# round 5, first GPU call: LDS-DMA hazard (stand-alone + in the factorisation), Gram lab, bench line with the new accounting
set -o pipefail
OUT=gpurun_out/r5a
mkdir -p $OUT
timeout -k 10 120 tools/ldsdma/ldsdma_hazard 512 200 > $OUT/ldsdma_hazard.txt 2>&1; echo "hazard rc=$?"
cat $OUT/ldsdma_hazard.txt
for v in 0 1 2 3; do
  echo "== MRBF_MEGA_PANELDMA=$v" >> $OUT/paneldma.txt
  MRBF_MEGA_PANELDMA=$v timeout -k 10 300 python3 tools/mega_check.py 1024,2048,4096,8192 3 3 >> $OUT/paneldma.txt 2>&1 || echo "paneldma $v failed rc=$?"
done
cat $OUT/paneldma.txt
timeout -k 10 300 tools/gramlab/gramlab 8192 8192 > $OUT/gramlab_ld8192.txt 2>&1; echo "gramlab rc=$?"
cat $OUT/gramlab_ld8192.txt
timeout -k 10 300 tools/gramlab/gramlab 8192 8320 > $OUT/gramlab_ld8320.txt 2>&1; echo "gramlab rc=$?"
cat $OUT/gramlab_ld8320.txt
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench rc=$?"
tail -c 3000 $OUT/bench_c3.json
