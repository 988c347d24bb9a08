"""Developer probe: the fused scoring + top-k kernel alone (for counter passes and quick timings).
python tools/eval_probe.py [--model biased|random] [--mode bf16|f32] [--users N] [--I items] [--d factors] [--k K] [--reps R]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device, SCORE_BF16, SCORE_F32  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="biased", choices=["biased", "random"])
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--users", type=int, default=262144)
    ap.add_argument("--U", type=int, default=1_000_000)
    ap.add_argument("--I", type=int, default=200_000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cuda = torch.device("cuda:0")
    gen = torch.Generator(device=cuda); gen.manual_seed(2018)
    stream = torch.cuda.Stream(device=cuda)
    torch.cuda.set_stream(stream)
    dev = Device(0, stream=stream.cuda_stream)
    P = (torch.rand(a.U, a.d, device=cuda, generator=gen) - 0.5) * 0.1
    Q = (torch.rand(a.I, a.d, device=cuda, generator=gen) - 0.5) * 0.1
    # "biased": item biases dominate the scores, as after training on popularity-skewed data
    b = torch.randn(a.I, device=cuda, generator=gen) if a.model == "biased" else torch.zeros(a.I, device=cuda)
    n = a.users
    eu = torch.arange(n, device=cuda, dtype=torch.int32)
    deg = 20
    indptr = torch.arange(0, deg * (a.U + 1), deg, device=cuda, dtype=torch.int64)
    indices = torch.sort(torch.randint(0, a.I, (a.U, deg), device=cuda, generator=gen, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    ids = torch.empty(n, a.k, device=cuda, dtype=torch.int32); sc = torch.empty(n, a.k, device=cuda)
    mode = SCORE_BF16 if a.mode == "bf16" else SCORE_F32
    dev.score_topk(P, Q, b, eu, a.k, train=(indptr, indices), mode=mode, out_ids=ids, out_scores=sc)
    torch.cuda.synchronize()
    import ctypes
    prof = getattr(dev.lib, "tfrec_debug_prof", None) if hasattr(dev.lib, "tfrec_debug_prof") else None
    buf = (ctypes.c_ulonglong * 16)()
    if prof is not None:
        prof(buf, 1)
    for _ in range(a.reps):
        t0 = time.perf_counter()
        dev.score_topk(P, Q, b, eu, a.k, train=(indptr, indices), mode=mode, out_ids=ids, out_scores=sc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fl = 2.0 * a.I * a.d * n / dt
        if prof is not None:
            prof(buf, 1)
            w = max(buf[7], 1)
            names = ["total", "filter+insert", "compaction", "store+barrier", "passing tiles", "compactions", "final", "waves", "c.fence", "c.load", "c.check", "c.descent", "c.write", "descent rounds", "id phases / tile store", "tile load wait"]
            print("  per wave: " + ", ".join(f"{nm}={buf[i] / w:,.0f}" for i, nm in enumerate(names)), flush=True)
        print(f"eval {a.mode} {a.model}: {n} users x {a.I} items d={a.d} k={a.k}: {dt*1e3:.2f} ms  {n/dt:,.0f} users/s  {fl/1e12:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
