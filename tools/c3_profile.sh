ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_c; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_c3" -o c3 -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-callers > "$OUT/bench_c3_under_profiler.json" 2> "$OUT/prof_c3.err"; cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof_c3/c3_results.db $OUT/kernel_stats.csv 14
find $OUT -name "*.db" -size +30M -delete
