#!/usr/bin/env python3
"""Device micro-benchmarks: sustained f64 MFMA rate, rocBLAS dgemm reference, built-in vs rocSOLVER potrf."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg  # noqa: E402
from morbit.jl_amd import _lib  # noqa: E402

ctx = pkg.Context()
out = {}
ms, tf = ctypes.c_float(), ctypes.c_double()
for bpc, thr, it in ((1, 256, 4000), (2, 256, 4000), (4, 256, 4000), (1, 256, -4000), (2, 256, -4000), (1, 256, 40000)):
    ctx.check(ctx.lib.mrbf_debug_mfma_peak(ctx.h, bpc, thr, it, ctypes.byref(ms), ctypes.byref(tf)))
    out["mfma_f64_peak_bpc%d_thr%d_it%d" % (bpc, thr, it)] = dict(ms=ms.value, tflops=tf.value); print(bpc, thr, it, tf.value, flush=True)
G = np.random.default_rng(0).standard_normal((128, 160))
A128 = np.asfortranarray(G @ G.T / 160 + np.eye(128))
cyc, rt = ctypes.c_double(), ctypes.c_double()
for reps, impl in ((64, 1), (-64, 1), (-64, 0)):
    ctx.set_option(_lib.OPT_DIAG_IMPL, impl)
    ctx.check(ctx.lib.mrbf_debug_diag(ctx.h, _lib.as_ptr(A128), reps, ctypes.byref(ms), ctypes.byref(cyc), ctypes.byref(rt)))
    out["diag_reps%d" % reps] = dict(us_per_call=ms.value * 1e3, shader_cycles=cyc.value, realtime_us=rt.value)
    print("diag impl", impl, "reps", reps, "us/call", ms.value * 1e3, "cycles", cyc.value, "realtime us", rt.value, "clock GHz", cyc.value / max(rt.value, 1e-9) / 1e3, flush=True)
cpm = ctypes.c_double()
for variant in (0, 1, 2, 3):
    for bpc in (1, 2):
        ctx.check(ctx.lib.mrbf_debug_mfma_asm(ctx.h, variant, bpc, 2000, ctypes.byref(ms), ctypes.byref(tf), ctypes.byref(cpm)))
        out["mfma_asm_v%d_bpc%d" % (variant, bpc)] = dict(ms=ms.value, tflops=tf.value, cycles_per_mfma=cpm.value)
        print("asm variant", variant, "bpc", bpc, tf.value, cpm.value, flush=True)
if "--potrf" not in sys.argv:
    DG = ()
else:
    DG = ((8192, 8192, 8192), (8192, 8192, 256), (8192, 8192, 128))
for (m, n, k) in DG:
    ctx.check(ctx.lib.mrbf_debug_dgemm(ctx.h, m, n, k, ctypes.byref(ms), ctypes.byref(tf)))
    out["rocblas_dgemm_%dx%dx%d" % (m, n, k)] = dict(ms=ms.value, tflops=tf.value); print(m, n, k, tf.value, flush=True)
if "--potrf-run" in sys.argv:
    for n in (2048, 4096, 8192):
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, 64))
        A = np.asfortranarray(G @ G.T / 64 + np.eye(n) * 4)
        for impl in (1, 2):
            F = A.copy(order="F")
            info = ctypes.c_int32()
            ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), impl, ctypes.byref(info), ctypes.byref(ms)))
            ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(A.copy(order="F")), impl, ctypes.byref(info), ctypes.byref(ms)))
            out["potrf_impl%d_n%d" % (impl, n)] = dict(ms=ms.value, tflops=n ** 3 / 3 / (ms.value * 1e-3) / 1e12, info=info.value)
print(json.dumps(out, indent=1))
