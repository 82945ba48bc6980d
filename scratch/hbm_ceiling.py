"""Measured ceilings of this box's HBM for plain streaming patterns (torch kernels): pure write, pure read, copy."""
import torch
dev = "cuda"
n = 1 << 30  # 1 GiB per buffer; rotate over 3 buffers to stay out of the 256 MB Infinity Cache
bufs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]


def timed(fn, reps=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


t = timed(lambda i: bufs[i % 4].fill_(1))
print("fill_  (write 1 GiB): %.1f us  %.2f TB/s" % (t * 1e6, n / t / 1e12))
f32 = [b.view(torch.float32) for b in bufs]
t = timed(lambda i: f32[i % 4].sum())
print("sum    (read 1 GiB):  %.1f us  %.2f TB/s" % (t * 1e6, n / t / 1e12))
t = timed(lambda i: bufs[(i + 1) % 4].copy_(bufs[i % 4]))
print("copy_  (read 1 + write 1 GiB): %.1f us  %.2f TB/s total" % (t * 1e6, 2 * n / t / 1e12))
h = [b.view(torch.float16) for b in bufs]
t = timed(lambda i: torch.add(h[i % 4][: n // 4], h[(i + 1) % 4][: n // 4], out=h[(i + 2) % 4][: n // 4]))
print("add f16 (read 2 x 0.5 + write 0.5 GiB): %.1f us  %.2f TB/s total" % (t * 1e6, 1.5 * n / t / 1e12))
