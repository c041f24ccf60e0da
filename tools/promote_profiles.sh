#!/bin/bash
# Turns what tools/collect_profiles.sh <tag> left under gpurun_out/<tag>/ into the summaries committed under profiles/ (prefix <tag>_).
#   tools/promote_profiles.sh r05_c
set -e
TAG=${1:?tag}
IN=gpurun_out/$TAG
P=profiles
cp $IN/bench_c3.json $P/${TAG}_bench.json
for c in c2 c4 c5; do cp $IN/bench_$c.json $P/${TAG}_bench_$c.json; done
cp $IN/bench_c3_under_profiler.json $P/${TAG}_bench_c3_under_profiler.json
# invocations per database: the cycles (C3: 10 + 3 warm-up + the cpu-baseline-free side launches; see profiles/README.md)
python3 tools/profile_summary.py stats $IN/prof_c3/c3_results.db $P/${TAG}_kernel_stats.csv 14
python3 tools/profile_summary.py stats $IN/prof_c2/c2_results.db $P/${TAG}_c2_kernel_stats.csv 24
python3 tools/profile_summary.py stats $IN/prof_c4/c4_results.db $P/${TAG}_c4_kernel_stats.csv 6
python3 tools/profile_summary.py stats $IN/prof_c5/c5_results.db $P/${TAG}_c5_kernel_stats.csv 7
python3 tools/profile_summary.py stats $IN/prof_r4/r4_results.db $P/${TAG}_round4_d64_kernel_stats.csv 3
python3 tools/profile_summary.py stats $IN/prof_r4_d128/r4_results.db $P/${TAG}_round4_d128_kernel_stats.csv 3
python3 tools/profile_summary.py pmc $IN/pmc_w/w_results.db $IN/pmc_f/f_results.db $P/${TAG}_pmc 5 C3
python3 tools/profile_summary.py pmc $IN/pmc_w_c2/w_results.db $IN/pmc_f_c2/f_results.db $P/${TAG}_c2_pmc 5 C2
python3 tools/profile_summary.py pmc $IN/pmc_w_c4/w_results.db $IN/pmc_f_c4/f_results.db $P/${TAG}_c4_pmc 360 C4
python3 tools/profile_summary.py pmc $IN/pmc_w_c5/w_results.db $IN/pmc_f_c5/f_results.db $P/${TAG}_c5_pmc 5 C5
{ echo "# tools/round4_bench.py (wall times of mrbf_round4 incl. the upload of the candidates; three repetitions per shape)"; cat $IN/round4_d64.txt $IN/round4_d128.txt $IN/round4_d24.txt; } > $P/${TAG}_round4_timing.txt   # (check the output for tracebacks before committing it)
{ echo "# per-launch timeline of the last round-4 call under rocprofv3 (tools/r4_timeline.py): start / duration in us, grid, queue (q1 main, the other the side stream)"; echo "## d = 64, 10^4 candidates"; cat $IN/round4_timeline_d64.txt; echo "## d = 128, 6000 candidates"; cat $IN/round4_timeline_d128.txt; } > $P/${TAG}_round4_timeline.txt
cp $IN/walklab.txt $P/${TAG}_walklab.txt
{ cat $IN/ps_step.txt; cat $IN/ps_step_d12.txt 2>/dev/null || true; } > $P/${TAG}_ps_step_timing.txt
for f in typical_latency iteration_c1 iteration_c4; do [ -f $IN/$f.txt ] && grep -v amdgpu.ids $IN/$f.txt > $P/${TAG}_$f.txt; done
ls $P | grep "^${TAG}_"
