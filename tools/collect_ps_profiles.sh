#!/bin/bash
# What changed in the second half of round 6 (PS ranking / sweeps): the default bench line with its callers, PS step timings, kernel
# statistics of the step at the three BASELINE dimensions, one-ranking timings, the iteration tables.   tools/collect_ps_profiles.sh <tag>
TAG=${1:-r06_c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
echo "bench done"
python3 tools/ps_bench2.py 64,128,256 > "$OUT/ps_step.txt" 2>&1
python3 tools/ps_bench.py > "$OUT/ps_step_d12.txt" 2>&1 || true
python3 tools/ps_rank_time.py > "$OUT/ps_rank_timing.txt" 2>&1 || true
for d in 64 128 256; do
  cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_ps$d" -o ps -- python3 "$ROOT/tools/ps_bench2.py" $d > "$OUT/ps${d}_prof.txt" 2>&1; cd "$ROOT"
  python3 tools/profile_summary.py stats "$OUT/prof_ps$d/ps_results.db" "$OUT/ps${d}_kernel_stats.csv" 3
done
echo "ps traces done"
python3 tools/iteration_latency.py C1 20 > "$OUT/iteration_c1.txt" 2>&1 || true
python3 tools/iteration_latency.py C4 20 > "$OUT/iteration_c4.txt" 2>&1 || true
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -size +30M -delete
ls "$OUT"
