#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# job log + chain trace of one persistent factorisation of the C3 matrix (n = 8192) -> gpurun_out/<tag>_{jlog,trace}.txt and their summaries
TAG=${1:-mp}; N=${2:-8192}
cat > gpurun_out/mp_run.py <<PY
import importlib, numpy as np, os, sys
pkg = importlib.import_module("morbit.jl_amd")
rng = np.random.default_rng(3)
n, d = $N, 64
C = rng.random((n, d)); Y = np.stack([((C-1)**2).sum(1), ((C+1)**2).sum(1)], 1) / d
cfg = pkg.RbfConfig(kernel="multiquadric")
for i in range(3):
    m = pkg.update_model(cfg, C, Y); print("ms_factor", m.info["ms_factor"], m.info.get("ms_factor_device")); m.free()
PY
PYTHONPATH=$(pwd) python3 gpurun_out/mp_run.py > gpurun_out/${TAG}_plain.txt 2>&1
MRBF_MEGA_JLOG=gpurun_out/${TAG}_jlog.txt PYTHONPATH=$(pwd) python3 gpurun_out/mp_run.py > /dev/null 2>&1
MRBF_MEGA_TRACE=gpurun_out/${TAG}_trace.txt PYTHONPATH=$(pwd) python3 gpurun_out/mp_run.py > /dev/null 2>&1
python3 tools/mega_jlog.py gpurun_out/${TAG}_jlog.txt > gpurun_out/${TAG}_jlog_summary.txt 2>&1
python3 tools/mega_trace.py gpurun_out/${TAG}_trace.txt > gpurun_out/${TAG}_trace_summary.txt 2>&1
cat gpurun_out/${TAG}_plain.txt; head -70 gpurun_out/${TAG}_jlog_summary.txt; tail -12 gpurun_out/${TAG}_trace_summary.txt
