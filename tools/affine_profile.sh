ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6q; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/affine_bench.py 128 300 > $OUT/plain.txt 2>&1
MRBF_EXPERIMENTS=1 MRBF_AFFINE_FUSED=0 python3 tools/affine_bench.py 128 300 >> $OUT/plain.txt 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o a -- python3 $ROOT/tools/affine_bench.py 128 300 > $OUT/prof.txt 2>&1; cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof/a_results.db $OUT/kernel_stats.csv 5
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/plain.txt; python3 - <<'PY'
import csv
for r in list(csv.reader(open('gpurun_out/r6q/kernel_stats.csv')))[1:8]:
    print("%-60s calls/call %7s avg_us %9.1f ms/call %7s" % (r[0][:60], r[7], float(r[3])/1e3, r[8]))
PY
