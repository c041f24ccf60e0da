ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6r; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/round4_bench.py 128 1700 2>&1 | grep -v amdgpu
python3 tools/round4_bench.py 128 600 2>&1 | grep -v amdgpu
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o r4 -- python3 $ROOT/tools/round4_bench.py 128 1700 > $OUT/prof.txt 2>&1; cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof/r4_results.db $OUT/kernel_stats.csv 3
