ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6c4; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o c4 -- python3 $ROOT/bench.py --config C4 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt; cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof/c4_results.db $OUT/kernel_stats.csv 6
