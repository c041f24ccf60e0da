#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5d
mkdir -p $OUT
rm -f gpurun_out/ps_omega_ratios.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"
tail -5 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
cat gpurun_out/ps_omega_ratios.txt
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$OUT/bench_c3.json'));print('C3',d['value'],d['phases_ms']);print({k:(round(v['frac'],3),round(v['ms'],4)) for k,v in d['kernels'].items()})"
python3 bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 -c "
import json;d=json.load(open('$OUT/bench_c2.json'));print('C2',d['value'],d['phases_ms'])"
