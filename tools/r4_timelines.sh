ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_c; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_r4 -o r4 -- python3 $ROOT/tools/round4_bench.py 64 10000 > $OUT/round4_under_profiler.txt 2>&1; cd $ROOT
python3 tools/r4_timeline.py $OUT/prof_r4/r4_results.db 100 160 > $OUT/round4_timeline_d64.txt 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_r4_d128 -o r4 -- python3 $ROOT/tools/round4_bench.py 128 6000 > $OUT/round4_d128_under_profiler.txt 2>&1; cd $ROOT
python3 tools/r4_timeline.py $OUT/prof_r4_d128/r4_results.db 300 360 > $OUT/round4_timeline_d128.txt 2>&1
ls -la $OUT/prof_r4/ $OUT/prof_r4_d128/
find $OUT -name "*.db" -size +30M -delete
