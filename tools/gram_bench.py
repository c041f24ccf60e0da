#!/usr/bin/env python3
"""Gram-assembly micro-benchmark: kernel-only time (hipEvents inside mrbf_gram) and HBM roofline fraction."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg  # noqa: E402
from morbit.jl_amd import _lib  # noqa: E402

ctx = pkg.Context()
for (n, d, kernel) in ((8192, 16, "multiquadric"), (8192, 16, "cubic"), (8192, 128, "multiquadric"), (8192, 64, "multiquadric"), (8192, 64, "gaussian"), (2048, 32, "gaussian"), (16384, 256, "cubic")):
    C = torch.rand((n, d), dtype=torch.float64, device="cuda")
    Phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, pkg.RbfConfig(kernel=kernel))
    ms = ctypes.c_float()
    best = 1e9
    for rep in range(6):
        ctx.check(ctx.lib.mrbf_gram(ctx.h, n, d, _lib.as_ptr(C), kid, a, b, 1, _lib.as_ptr(Phi), None, ctypes.byref(ms)))
        best = min(best, ms.value)
    gb = (8.0 * n * n + 8.0 * n * d) / 1e9
    print("gram n=%d d=%d %-13s %.3f ms  %.0f GB/s  frac of 8 TB/s = %.3f" % (n, d, kernel, best, gb / (best * 1e-3), gb / (best * 1e-3) / 8000), flush=True)
