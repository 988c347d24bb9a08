"""Developer probe: the APR step at the reference's batch size (128 triples)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device

torch.cuda.set_device(0)
cuda = torch.device("cuda:0")
stream = torch.cuda.Stream(device=cuda)
torch.cuda.set_stream(stream)
dev = Device(0, stream=stream.cuda_stream)
U, I, d = 943, 1682, 64
g = torch.Generator(device=cuda); g.manual_seed(1)
P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.1
Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.1
b = torch.zeros(I, device=cuda)
acc = [torch.full_like(x, 0.1) for x in (P, Q, b)]
dl = [torch.zeros_like(x) for x in (P, Q, b)]
for B in (128, 512):
    n = 256
    u = torch.randint(0, U, (n, B), device=cuda, generator=g, dtype=torch.int32)
    i = torch.randint(0, I, (n, B), device=cuda, generator=g, dtype=torch.int32)
    j = torch.randint(0, I, (n, B), device=cuda, generator=g, dtype=torch.int32)

    def step(s):
        dev.apr_adagrad_step(P, Q, b, acc[0], acc[1], acc[2], dl[0], dl[1], dl[2], u[s], i[s], j[s], 0.05, 0.0, 1.0, 0.5)
    for s in range(20):
        step(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for s in range(n):
        step(s)
    e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"APR B={B}: host {th / n * 1e6:.1f} us/call, stream {e0.elapsed_time(e1) / n * 1e3:.1f} us/step", flush=True)
    dev.set_profiling(True)
    step(0)
    print("   phases us:", {k: round(v * 1e3, 1) for k, v in dev.phase_times()}, flush=True)
    dev.set_profiling(False)
