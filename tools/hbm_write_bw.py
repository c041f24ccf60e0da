"""Achievable HBM write (and copy) bandwidth on this GPU for a buffer of the Gram matrix's size (8 n^2 bytes, n = 8192)."""
import torch
n = 8192
x = torch.empty(n * n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(1.5)); print(f"fill_  {x.numel()*8/1e6:.0f} MB: {ms*1e3:.1f} us -> {x.numel()*8/ms/1e9:.2f} TB/s (write only)")
ms = t(lambda: x.zero_()); print(f"zero_ : {ms*1e3:.1f} us -> {x.numel()*8/ms/1e9:.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy_ : {ms*1e3:.1f} us -> {2*x.numel()*8/ms/1e9:.2f} TB/s (read + write)")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   : {ms*1e3:.1f} us -> {2*x.numel()*8/ms/1e9:.2f} TB/s (read + write)")
ms = t(lambda: x.sum()); print(f"sum   : {ms*1e3:.1f} us -> {x.numel()*8/ms/1e9:.2f} TB/s (read only)")
