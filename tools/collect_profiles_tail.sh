#!/bin/bash
# Second half of tools/collect_profiles.sh (the whole set no longer fits one 20-minute gpurun call): round-4 traces / timelines, decision-kernel lab,
# PS-step timings, typical latencies with the CPU column, the iteration tables.   tools/collect_profiles_tail.sh <tag>
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_r4_d128" -o r4 -- python3 "$ROOT/tools/round4_bench.py" 128 6000 > "$OUT/round4_d128_under_profiler.txt" 2>&1; cd "$ROOT"
python3 tools/r4_timeline.py "$OUT/prof_r4/r4_results.db" 100 160 > "$OUT/round4_timeline_d64.txt" 2>&1 || true
python3 tools/r4_timeline.py "$OUT/prof_r4_d128/r4_results.db" 300 360 > "$OUT/round4_timeline_d128.txt" 2>&1 || true
./tools/walklab/walklab 65 2 5 > "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 65 8 5 >> "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 129 8 5 >> "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 25 2 5 >> "$OUT/walklab.txt" 2>&1 || true
python3 tools/ps_bench2.py 64,128,256 > "$OUT/ps_step.txt" 2>&1
python3 tools/ps_bench.py > "$OUT/ps_step_d12.txt" 2>&1 || true
python3 tools/typical_latency.py > "$OUT/typical_latency.txt" 2>&1 || true
python3 tools/iteration_latency.py C1 20 > "$OUT/iteration_c1.txt" 2>&1 || true
python3 tools/iteration_latency.py C4 20 > "$OUT/iteration_c4.txt" 2>&1 || true
find "$OUT" -name "*.csv" -size +2M -delete
ls "$OUT"
