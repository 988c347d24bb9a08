"""Developer probe: known-byte-count kernels for calibrating the L2 memory-side request counters.
Run under `rocprofv3 --pmc ... --output-format csv`; prints the byte counts to compare against."""
import torch

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
n = 1 << 28                                   # 1 GiB of float32
x = torch.empty(n, device=dev, dtype=torch.float32).uniform_()
y = torch.empty_like(x)
rows = 1 << 24                                # 4 GiB table of 256-byte rows, far beyond every cache
table = torch.empty(rows, 64, device=dev, dtype=torch.float32).uniform_()
idx = torch.randint(0, rows, (1 << 21,), device=dev)
out = torch.empty(idx.numel(), 64, device=dev, dtype=torch.float32)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)                                # streaming: reads 1 GiB, writes 1 GiB
    torch.index_select(table, 0, idx, out=out)  # gathers 2^21 random 256-byte rows: reads 512 MiB, writes 512 MiB
    s = x.sum()                               # streaming read of 1 GiB
torch.cuda.synchronize()
print("copy: read %d write %d; gather: read %d write %d; sum: read %d" % (4 * n, 4 * n, idx.numel() * 256, idx.numel() * 256, 4 * n))
