"""Developer probe: where a 512-triple step spends its time (phases on the GPU, host time per call)."""
import os, sys, time
import numpy as np
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
for B in (128, 512, 4096):
    n = 256
    u = torch.randint(0, U, (n, B), device=cuda, generator=g, dtype=torch.int32)
    i = torch.randint(0, I, (n, B), device=cuda, generator=g, dtype=torch.int32)
    j = torch.randint(0, I, (n, B), device=cuda, generator=g, dtype=torch.int32)
    rows = [(u[s], i[s], j[s]) for s in range(n)]
    for name, fn in (("plain", dev.bpr_sgd_step), ("noalloc", dev.bpr_sgd_step_noalloc)):
        for s in range(20):
            fn(P, Q, b, *rows[s], 0.01, 0.05)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for s in range(n):
            fn(P, Q, b, *rows[s], 0.01, 0.05)
        e1.record(); t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"B={B} {name}: host {t_host / n * 1e6:.1f} us/call, stream {e0.elapsed_time(e1) / n * 1e3:.1f} us/step", flush=True)
    dev.set_profiling(True)
    acc = {}
    for s in range(20):
        dev.bpr_sgd_step(P, Q, b, *rows[s], 0.01, 0.05)
        for k, v in dev.phase_times():
            acc[k] = acc.get(k, 0) + v / 20
    dev.set_profiling(False)
    print("   phases us:", {k: round(v * 1e3, 1) for k, v in acc.items()}, flush=True)
