# per-kernel statistics of the PS step under rocprofv3 (tools/ps_bench2.py runs three steps per dimension); usage: bash tools/ps_profile.sh [dims...]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6p; mkdir -p $OUT; export TMPDIR=/tmp
DIMS=${@:-64 128}
for d in $DIMS; do
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_ps$d -o ps -- python3 $ROOT/tools/ps_bench2.py $d > $OUT/ps${d}_prof.txt 2>&1; cd $ROOT
python3 tools/profile_summary.py stats $OUT/prof_ps$d/ps_results.db $OUT/ps${d}_kernel_stats.csv 3
done
find $OUT -name "*.csv" -size +2M -delete
for d in $DIMS; do echo "== d=$d"; grep -v amdgpu $OUT/ps${d}_prof.txt | tail -3; cut -c1-110 $OUT/ps${d}_kernel_stats.csv | awk -F, '{print $1","$(NF-5)","$(NF-1)","$NF}' | head -14; done
