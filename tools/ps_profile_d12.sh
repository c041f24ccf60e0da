ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6p12; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o ps -- python3 $ROOT/tools/ps_bench.py > $OUT/prof.txt 2>&1; cd $ROOT
echo done
