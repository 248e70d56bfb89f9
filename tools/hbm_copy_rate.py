"""Achievable HBM rate of this box with plain torch kernels (copy / in-place scale / read-only sum): the practical ceiling the KDyn
passes are compared with in DESIGN.md section 5."""
import torch, time
for mb in (64, 184, 512, 2048):
    n = mb * 1024 * 1024 // 8
    a = torch.randn(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print("copy %5d MB: %.1f us, %.2f TB/s (read+write)" % (mb, t * 1e6, 2 * n * 8 / t / 1e12))
    for _ in range(3): a.mul_(1.0001)
    e0.record()
    for _ in range(20): a.mul_(1.0001)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print("scale %5d MB in place: %.1f us, %.2f TB/s" % (mb, t * 1e6, 2 * n * 8 / t / 1e12))
    s = a.sum()
    e0.record()
    for _ in range(20): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print("read-only sum %5d MB: %.1f us, %.2f TB/s" % (mb, t * 1e6, n * 8 / t / 1e12))
