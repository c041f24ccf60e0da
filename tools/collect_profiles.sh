#!/bin/bash
# One GPU-box call that gathers what profiles/ is built from: bench lines of the four BASELINE configurations, rocprofv3 kernel
# stats of C3 / C4 / C5, the two PMC passes of C3 (separate runs, --kernel-trace only), round-4 and PS-step timings.
#   tools/collect_profiles.sh <tag>        (outputs under gpurun_out/<tag>/; summaries promoted with tools/profile_summary.py)
set -e
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
echo "C3 done"
python3 bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
echo "bench lines done"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_c3" -o c3 -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-callers > "$OUT/bench_c3_under_profiler.json" 2> "$OUT/prof_c3.err"
echo "C3 trace done"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_c4" -o c4 -- python3 "$ROOT/bench.py" --config C4 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4_under_profiler.json" 2> "$OUT/prof_c4.err"
echo "C4 trace done"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_c5" -o c5 -- python3 "$ROOT/bench.py" --config C5 --steps 2 --warmup 1 --problems 2 --no-cpu-baseline > "$OUT/bench_c5_under_profiler.json" 2> "$OUT/prof_c5.err"
echo "C5 trace done"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_c2" -o c2 -- python3 "$ROOT/bench.py" --config C2 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_c2_under_profiler.json" 2> "$OUT/prof_c2.err"
echo "C2 trace done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_w" -o w -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-callers > "$OUT/pmc_w.json" 2> "$OUT/pmc_w.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_f" -o f -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-callers > "$OUT/pmc_f.json" 2> "$OUT/pmc_f.err"
echo "C3 PMC passes done"
# PMC passes of the other configurations (roofline.traffic must never be null): C2 (20 cycles), C4 (4 batches), C5 (2 problems x 2 steps)
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_w_c2" -o w -- python3 "$ROOT/bench.py" --config C2 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_w_c2.json" 2> "$OUT/pmc_w_c2.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_f_c2" -o f -- python3 "$ROOT/bench.py" --config C2 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_f_c2.json" 2> "$OUT/pmc_f_c2.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_w_c4" -o w -- python3 "$ROOT/bench.py" --config C4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_w_c4.json" 2> "$OUT/pmc_w_c4.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_f_c4" -o f -- python3 "$ROOT/bench.py" --config C4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_f_c4.json" 2> "$OUT/pmc_f_c4.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_w_c5" -o w -- python3 "$ROOT/bench.py" --config C5 --steps 1 --warmup 1 --problems 2 --no-cpu-baseline > "$OUT/pmc_w_c5.json" 2> "$OUT/pmc_w_c5.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_f_c5" -o f -- python3 "$ROOT/bench.py" --config C5 --steps 1 --warmup 1 --problems 2 --no-cpu-baseline > "$OUT/pmc_f_c5.json" 2> "$OUT/pmc_f_c5.err"
echo "PMC passes done"
cd "$ROOT"
python3 tools/round4_bench.py 64 10000 > "$OUT/round4_d64.txt" 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_r4" -o r4 -- python3 "$ROOT/tools/round4_bench.py" 64 10000 > "$OUT/round4_under_profiler.txt" 2>&1; cd "$ROOT"
python3 tools/round4_bench.py 24 3000 > "$OUT/round4_d24.txt" 2>&1
python3 tools/round4_bench.py 128 6000 > "$OUT/round4_d128.txt" 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_r4_d128" -o r4 -- python3 "$ROOT/tools/round4_bench.py" 128 6000 > "$OUT/round4_d128_under_profiler.txt" 2>&1; cd "$ROOT"
python3 tools/r4_timeline.py "$OUT/prof_r4/r4_results.db" 100 160 > "$OUT/round4_timeline_d64.txt" 2>&1 || true
python3 tools/r4_timeline.py "$OUT/prof_r4_d128/r4_results.db" 300 360 > "$OUT/round4_timeline_d128.txt" 2>&1 || true
./tools/walklab/walklab 65 2 5 > "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 65 8 5 >> "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 129 8 5 >> "$OUT/walklab.txt" 2>&1 || true
./tools/walklab/walklab 25 2 5 >> "$OUT/walklab.txt" 2>&1 || true
python3 tools/ps_bench2.py 64,128,256 > "$OUT/ps_step.txt" 2>&1
python3 tools/typical_latency.py > "$OUT/typical_latency.txt" 2>&1 || true
python3 tools/iteration_latency.py C1 20 > "$OUT/iteration_c1.txt" 2>&1 || true
python3 tools/iteration_latency.py C4 20 > "$OUT/iteration_c4.txt" 2>&1 || true
python3 tools/ps_bench.py > "$OUT/ps_step_d12.txt" 2>&1 || true
ls "$OUT"
# the raw traces are large: keep the databases only
find "$OUT" -name "*.csv" -size +2M -delete
du -sh "$OUT"
