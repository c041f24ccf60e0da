export MRBF_EXPERIMENTS=1
mkdir -p gpurun_out/r6h
cat > gpurun_out/r6h/run.py <<PY
import importlib, numpy as np
pkg = importlib.import_module("morbit.jl_amd")
rng = np.random.default_rng(3)
n, d = 2048, 32
C = rng.random((n, d)); Y = (C**2).sum(1, keepdims=True)
cfg = pkg.RbfConfig(kernel="gaussian")
for i in range(3):
    m = pkg.update_model(cfg, C, Y); print("ms_factor", m.info["ms_factor"], m.info.get("ms_factor_device")); m.free()
PY
PYTHONPATH=$(pwd) python3 gpurun_out/r6h/run.py > gpurun_out/r6h/plain.txt 2>&1
MRBF_MEGA_TRACE=gpurun_out/r6h/trace.txt MRBF_MEGA_JLOG=gpurun_out/r6h/jlog.txt PYTHONPATH=$(pwd) python3 gpurun_out/r6h/run.py > gpurun_out/r6h/traced.txt 2>&1
python3 tools/mega_gap.py gpurun_out/r6h/trace.txt gpurun_out/r6h/jlog.txt > gpurun_out/r6h/gap.txt 2>&1
cat gpurun_out/r6h/plain.txt gpurun_out/r6h/traced.txt; cat gpurun_out/r6h/gap.txt
