"""Ceiling for a write-only kernel on this box: torch's fill / zero kernels and hipMemsetAsync over 1.07 GB (the lower triangle of the
N = 16 384 fp64 matrix is 1.074 GB), beside the assembly's own rate (bench.py roofline_secondary)."""
import torch

n = 1073807360 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda")
for name, fn in (("fill_(1.5)", lambda: x.fill_(1.5)), ("zero_()", lambda: x.zero_()),
                 ("copy_(y) [read + write]", None)):
    if fn is None:
        y = torch.empty_like(x)
        fn = lambda: x.copy_(y)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:28s} {ms:7.4f} ms  {n * 8 / ms / 1e9:7.1f} GB/s written")
