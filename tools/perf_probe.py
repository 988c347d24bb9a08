"""Developer probe: BPR step phase times and evaluation rates on one GPU (not part of the contract)."""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device, SCORE_BF16, SCORE_F32  # noqa: E402


def zipf_items(n_items, size, gen, device, alpha=1.0):
    w = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float64) ** alpha
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    r = torch.rand(size, device=device, dtype=torch.float64, generator=gen)
    ranks = torch.searchsorted(cdf, r).clamp_(max=n_items - 1)
    perm = torch.randperm(n_items, device=device, generator=gen)
    return perm[ranks].to(torch.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--U", type=int, default=1_000_000)
    ap.add_argument("--I", type=int, default=200_000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--batches", type=str, default="65536,262144,1048576")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dist", type=str, default="zipf")
    ap.add_argument("--eval-users", type=int, default=0)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--apr", type=int, default=0, help="time the APR step at this batch size")
    ap.add_argument("--dns", type=int, default=0, help="time the DNS selection for this many positives (32 candidates each)")
    ap.add_argument("--versioned", action="store_true", help="steps on a versioned user table (tfrec_utable)")
    ap.add_argument("--plain-only", action="store_true", help="only the un-pipelined steps (for counter collection)")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cuda = torch.device("cuda:0")
    gen = torch.Generator(device=cuda); gen.manual_seed(2018)
    stream = torch.cuda.Stream(device=cuda)
    torch.cuda.set_stream(stream)
    dev = Device(0, stream=stream.cuda_stream)
    P = (torch.rand(a.U, a.d, device=cuda, generator=gen) - 0.5) * 0.1
    Q = (torch.rand(a.I, a.d, device=cuda, generator=gen) - 0.5) * 0.1
    b = torch.zeros(a.I, device=cuda)
    ut = dev.user_table(P) if a.versioned else None
    for B in [int(x) for x in a.batches.split(",") if x]:
        n = B * 4
        users = torch.randint(0, a.U, (n,), device=cuda, generator=gen, dtype=torch.int32)
        pos = zipf_items(a.I, n, gen, cuda) if a.dist == "zipf" else torch.randint(0, a.I, (n,), device=cuda, generator=gen, dtype=torch.int32)
        neg = torch.randint(0, a.I, (n,), device=cuda, generator=gen, dtype=torch.int32)
        dev.reserve(B, a.d)
        def step(s):
            o = (s % 4) * B
            if ut is not None:
                dev.bpr_sgd_step_v(ut, Q, b, users[o:o + B], pos[o:o + B], neg[o:o + B], 0.01, 0.05)
            else:
                dev.bpr_sgd_step(P, Q, b, users[o:o + B], pos[o:o + B], neg[o:o + B], 0.01, 0.05)
        for s in range(3):
            step(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in range(a.steps):
            step(s)
        e1.record(); torch.cuda.synchronize()
        ms_plain = e0.elapsed_time(e1) / a.steps
        if a.plain_only:
            print(f"B={B}: plain {ms_plain:.3f} ms/step", flush=True)
            continue

        def batch(s):
            o = (s % 4) * B
            return users[o:o + B], pos[o:o + B], neg[o:o + B]
        for K in (1, 2, 4):
            grp = (users[:K * B], pos[:K * B], neg[:K * B])
            dev.prepare_batches(0, a.U, a.I, *grp, K)
            e0.record()
            nst = (a.steps // K) * K
            for s in range(nst):
                g, r = divmod(s, K)
                if r == 0:
                    dev.prepare_batches((g + 1) & 1, a.U, a.I, *grp, K)
                if ut is not None:
                    dev.bpr_sgd_step_prepared_v(g & 1, ut, Q, b, 0.01, 0.05)
                else:
                    dev.bpr_sgd_step_prepared(g & 1, P, Q, b, 0.01, 0.05)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / nst
            print(f"B={B}: plain {ms_plain:.3f} ms/step, pipelined K={K} {ms:.3f} ms/step")
        rate = B / ms * 1e3
        alg = (24 * a.d + 28) * rate / 1e12
        dev.set_profiling(True)
        acc = {}
        for s in range(5):
            step(s)
            for name, t in dev.phase_times():
                acc[name] = acc.get(name, 0.0) + t / 5
        dev.set_profiling(False)
        uu = torch.unique(users[:B]).numel(); ui = torch.unique(torch.cat([pos[:B], neg[:B]])).numel()
        print(f"B={B:8d} {a.dist}: {ms:8.3f} ms/step  {rate/1e9:6.3f} G triples/s  algorithmic {alg:5.2f} TB/s ({alg/8*100:4.1f}% of 8 TB/s)"
              f"  uniq users {uu} items {ui}  phases " + " ".join(f"{k}={v*1e3:.0f}us" for k, v in acc.items()), flush=True)
    if ut is not None:
        ut.close()
    if a.apr:
        B = a.apr
        users = torch.randint(0, a.U, (B,), device=cuda, generator=gen, dtype=torch.int32)
        pos = zipf_items(a.I, B, gen, cuda)
        neg = torch.randint(0, a.I, (B,), device=cuda, generator=gen, dtype=torch.int32)
        accP, accQ, accb = torch.full_like(P, 0.1), torch.full_like(Q, 0.1), torch.full_like(b, 0.1)
        dP, dQ, db = torch.zeros_like(P), torch.zeros_like(Q), torch.zeros_like(b)
        for s_ in range(3):
            dev.apr_adagrad_step(P, Q, b, accP, accQ, accb, dP, dQ, db, users, pos, neg, 0.05, 0.0, 1.0, 0.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in range(a.steps):
            dev.apr_adagrad_step(P, Q, b, accP, accQ, accb, dP, dQ, db, users, pos, neg, 0.05, 0.0, 1.0, 0.5)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        alg = (84 * a.d + 68) * B / ms * 1e3 / 1e12
        dev.set_profiling(True)
        dev.apr_adagrad_step(P, Q, b, accP, accQ, accb, dP, dQ, db, users, pos, neg, 0.05, 0.0, 1.0, 0.5)
        ph = " ".join(f"{k}={v*1e3:.0f}us" for k, v in dev.phase_times())
        dev.set_profiling(False)
        print(f"APR B={B} d={a.d}: {ms:.3f} ms/step  {B/ms*1e3/1e9:.3f} G triples/s  algorithmic {alg:.2f} TB/s ({alg/8*100:.1f}% of 8 TB/s)  {ph}", flush=True)
    if a.dns:
        n, Cn = a.dns, 32
        users = torch.randint(0, a.U, (n,), device=cuda, generator=gen, dtype=torch.int32)
        cand = torch.randint(0, a.I, (n, Cn), device=cuda, generator=gen, dtype=torch.int32)
        out = torch.empty(n, device=cuda, dtype=torch.int32)
        for s_ in range(3):
            dev.dns_select(P, Q, b, users, cand, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in range(a.steps):
            dev.dns_select(P, Q, b, users, cand, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        alg = ((Cn + 1) * 4 * a.d + 8 * Cn) * n / ms * 1e3 / 1e12
        print(f"DNS select n={n} C={Cn} d={a.d}: {ms:.3f} ms  {n/ms*1e3/1e9:.3f} G positives/s  algorithmic {alg:.2f} TB/s ({alg/8*100:.1f}% of 8 TB/s)", flush=True)
    if a.eval_users:
        n = a.eval_users
        eu = torch.arange(n, device=cuda, dtype=torch.int32)
        deg = 20
        indptr = torch.arange(0, deg * (a.U + 1), deg, device=cuda, dtype=torch.int64)
        indices = torch.sort(torch.randint(0, a.I, (a.U, deg), device=cuda, generator=gen, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
        ids = torch.empty(n, a.k, device=cuda, dtype=torch.int32); sc = torch.empty(n, a.k, device=cuda)
        for mode, name in ((SCORE_F32, "f32"), (SCORE_BF16, "bf16")):
            dev.score_topk(P, Q, b, eu, a.k, train=(indptr, indices), mode=mode, out_ids=ids, out_scores=sc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.score_topk(P, Q, b, eu, a.k, train=(indptr, indices), mode=mode, out_ids=ids, out_scores=sc)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            fl = 2.0 * a.I * a.d * n / dt
            print(f"eval {name}: {n} users x {a.I} items d={a.d} k={a.k}: {dt*1e3:.1f} ms  {n/dt:,.0f} users/s  {fl/1e12:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
