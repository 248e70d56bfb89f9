"""Does a kernel whose input was written by the launch before it run faster than the HBM rate?  Ping-pong copies a -> b -> a of S megabytes
(each launch reads what the previous one wrote; the pair's footprint is 2 S) against the same copy between cold 2-GB buffers, and a
three-buffer chain a -> b -> c -> a.  Plain torch kernels; read + written bytes per second.  (the 128^3 step hands 75-113 MB
from kernel to kernel; the cache behind the L2s holds 256 MB.)"""
import torch


def rate(bufs, reps=40):
    k = len(bufs)
    for i in range(2 * k): bufs[(i + 1) % k].copy_(bufs[i % k])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): bufs[(i + 1) % k].copy_(bufs[i % k])
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    return t * 1e6, 2 * bufs[0].numel() * 8 / t / 1e12


for mb in (16, 32, 64, 96, 113, 128, 192, 256, 512, 2048):
    n = mb * 1024 * 1024 // 8
    two = [torch.randn(n, dtype=torch.float64, device="cuda") for _ in range(2)]
    us2, r2 = rate(two)
    three = two + [torch.empty_like(two[0])]
    us3, r3 = rate(three)
    print("%5d MB  ping-pong %.1f us %.2f TB/s | chain of three %.1f us %.2f TB/s" % (mb, us2, r2, us3, r3), flush=True)
    del two, three
