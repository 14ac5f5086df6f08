"""Achievable HBM bandwidth of plain device copies and reads on this GPU (reference point for the edge kernels)."""
import torch
n = 1 << 30          # floats: 4 GiB per tensor
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: b.copy_(a))
print(f"copy 4 GiB: {ms:.3f} ms, {2 * 4 * n / ms / 1e9:.2f} TB/s (read + write)")
ms = timed(lambda: a.sum())
print(f"sum  4 GiB: {ms:.3f} ms, {4 * n / ms / 1e9:.2f} TB/s (read)")
ms = timed(lambda: torch.add(a, b, out=b))
print(f"add  4 GiB: {ms:.3f} ms, {3 * 4 * n / ms / 1e9:.2f} TB/s (2 reads + write)")
ms = timed(lambda: b.fill_(1.0))
print(f"fill 4 GiB: {ms:.3f} ms, {4 * n / ms / 1e9:.2f} TB/s (write)")
