"""Developer probe: what the memory system gives for the access patterns of the row passes
(random 256-byte row gathers, with and without row writes), using plain torch kernels."""
import torch

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
rows = 1 << 22                                 # 1 GiB table of 256-byte rows
table = torch.empty(rows, 64, device=dev).uniform_()
n = 1 << 21
idx = torch.randint(0, rows, (n,), device=dev)
out = torch.empty(n, 64, device=dev)
src = torch.empty(n, 64, device=dev).uniform_()
x = torch.empty(1 << 28, device=dev).uniform_()
y = torch.empty_like(x)


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timeit(lambda: torch.index_select(table, 0, idx, out=out))
print("gather 256B rows -> sequential rows: %.1f us, %.2f TB/s (read+write %d MB)" % (t * 1e6, 2 * n * 256 / t / 1e12, 2 * n * 256 >> 20))
t = timeit(lambda: table.index_copy_(0, idx, src))
print("sequential rows -> scatter 256B rows: %.1f us, %.2f TB/s" % (t * 1e6, 2 * n * 256 / t / 1e12))
t = timeit(lambda: y.copy_(x))
print("stream copy 1 GiB: %.1f us, %.2f TB/s" % (t * 1e6, 2 * x.numel() * 4 / t / 1e12))
t = timeit(lambda: x.sum())
print("stream read 1 GiB: %.1f us, %.2f TB/s" % (t * 1e6, x.numel() * 4 / t / 1e12))
t = timeit(lambda: y.zero_())
print("stream write 1 GiB: %.1f us, %.2f TB/s" % (t * 1e6, x.numel() * 4 / t / 1e12))
