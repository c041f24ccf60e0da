#!/bin/bash
# the round-4 part of tools/collect_profiles.sh alone: timings, kernel traces at both reference shapes, per-launch timelines, decision-kernel lab
#   tools/collect_round4.sh <tag>   (outputs under gpurun_out/<tag>/; promoted by tools/promote_profiles.sh)
set -e
TAG=${1:-r05_c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/round4_bench.py 64 10000 > "$OUT/round4_d64.txt" 2>&1
python3 tools/round4_bench.py 24 3000 > "$OUT/round4_d24.txt" 2>&1
python3 tools/round4_bench.py 128 6000 > "$OUT/round4_d128.txt" 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_r4" -o r4 -- python3 "$ROOT/tools/round4_bench.py" 64 10000 > "$OUT/round4_under_profiler.txt" 2>&1; cd "$ROOT"
cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_r4_d128" -o r4 -- python3 "$ROOT/tools/round4_bench.py" 128 6000 > "$OUT/round4_d128_under_profiler.txt" 2>&1; cd "$ROOT"
python3 tools/r4_timeline.py "$OUT/prof_r4/r4_results.db" 100 160 > "$OUT/round4_timeline_d64.txt" 2>&1 || true
python3 tools/r4_timeline.py "$OUT/prof_r4_d128/r4_results.db" 300 360 > "$OUT/round4_timeline_d128.txt" 2>&1 || true
./tools/walklab/walklab 65 2 5 > "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 65 8 5 >> "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 129 8 5 >> "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 25 2 5 >> "$OUT/walklab.txt" 2>&1 || true
find "$OUT" -name "*.csv" -size +2M -delete
tail -2 "$OUT/round4_d64.txt" | cut -c1-100; tail -1 "$OUT/round4_d128.txt" | cut -c1-100
