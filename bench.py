#!/usr/bin/env python3
"""Benchmark of the matrix-factorization ranking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one BPR-MF minibatch step (gather, dot, pairwise loss forward/backward, duplicate
reduction, SGD row updates) over one batch of synthetic triples that is already resident in
HBM.  N=1 runs the single-GPU configuration the metric is quoted on (1M users x 200k items,
d=64, fp32); N>1 (launched by torch.distributed.run, one rank per GPU) shards the user table by
rows; the item table (51 MB here) is replicated and the item-gradient sums are all-reduced once per
step (item tables above 256 MB are row-sharded too, with all-to-all lookups); weak scaling (the
per-GPU batch is fixed).  Rank 0 prints ONE JSON line.

Extra objects on that line: "roofline" (the step's kernels against the HBM roofline, timed with HIP
events on the stream the library launches on), "cpu_baseline" (the CPU oracle timed on this host,
rank 0, N=1 only), "eval" (full-rank evaluation users/s against the MFMA roofline, timed separately,
outside the K training steps, with its own "cpu_baseline": the reference's C++ evaluator on a bounded
user sample) and "other_paths" (APR step, DNS selection and the device epoch pipeline on the same tables).
"""
import argparse
import json
import os
import socket
import subprocess
import threading
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MEASURED_COPY_GBS = 6290.0
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}


def _sustained_bf16():
    """TFLOP/s that v_mfma_f32_32x32x16_bf16 sustained on random operands in tools/mfma_rate.hip (profiles/mfma_rate.json;
    measured once, not in the run: it takes a second of the whole chip)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "mfma_rate.json")) as f:
            return float(json.load(f)["sustained_bf16_32x32x16_random_tflops"])
    except (OSError, KeyError, ValueError):
        return None


SUSTAINED_BF16_TFLOPS = _sustained_bf16()


def zipf_items(n_items, size, gen, device, alpha=1.0):
    """Items ~ Zipf(alpha) over a random permutation of the ids (hot rows exist)."""
    w = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float64) ** alpha
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    out = torch.empty(size, device=device, dtype=torch.int32)
    perm = torch.randperm(n_items, device=device, generator=gen)
    chunk = 1 << 24
    for s in range(0, size, chunk):
        n = min(chunk, size - s)
        r = torch.rand(n, device=device, dtype=torch.float64, generator=gen)
        out[s:s + n] = perm[torch.searchsorted(cdf, r).clamp_(max=n_items - 1)].to(torch.int32)
    return out


def pmc_traffic(B, d, name="pmc_traffic.json"):
    """HBM bytes per step from the committed counter passes (profiles/pmc_traffic.json, written by
    tools/make_profile_summary.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same
    workload; pmc_traffic_large.json: the same for the 10M-user x 2M-item shape); None when the file does not
    describe this batch size and factor size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None
    if t.get("workload_batch") != B or t.get("factors") != d:
        return None
    return t["bytes_per_step"]


def other_paths(dev, P, Q, b, U, I, d, B, gen, cuda, steps=10, epoch_pipeline=True):
    """APR step (model/APR.py:156-164) and DNS selection (model/DNSBPR.py:68-79) on the same tables:
    throughput and the algorithmic HBM rate; timed outside the K steps of the headline metric."""
    out = {}
    users = torch.randint(0, U, (B,), device=cuda, generator=gen, dtype=torch.int32)
    pos = zipf_items(I, B, gen, cuda)
    neg = torch.randint(0, I, (B,), device=cuda, generator=gen, dtype=torch.int32)
    Pa, Qa, ba = P.clone(), Q.clone(), b.clone()
    acc = [torch.full_like(x, 0.1) for x in (Pa, Qa, ba)]
    dl = [torch.zeros_like(x) for x in (Pa, Qa, ba)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def apr():
        dev.apr_adagrad_step(Pa, Qa, ba, acc[0], acc[1], acc[2], dl[0], dl[1], dl[2], users, pos, neg, 0.05, 0.0, 1.0, 0.5)
    for _ in range(3):
        apr()
    e0.record()
    for _ in range(steps):
        apr()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    gbs = (84 * d + 68) * B / (ms * 1e-3) / 1e9
    out["apr_adagrad_step"] = {"triples_per_s": B / (ms * 1e-3), "ms_per_step": ms, "batch": B,
                               "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                            "algorithmic_bytes_per_triple": 84 * d + 68}}
    # the reference's own APR batch size (conf/APR.ini:4: 128 triples per step)
    small = 128
    for k in range(220):
        if k == 20:
            e0.record()
        o = (k % 1000) * small
        dev.apr_adagrad_step(Pa, Qa, ba, acc[0], acc[1], acc[2], dl[0], dl[1], dl[2], users[o:o + small], pos[o:o + small],
                             neg[o:o + small], 0.05, 0.0, 1.0, 0.5)
    e1.record(); torch.cuda.synchronize()
    out["apr_reference_batch_128"] = {"us_per_step": e0.elapsed_time(e1) * 1e3 / 200, "triples_per_s": 200 * small / (e0.elapsed_time(e1) * 1e-3)}
    del acc, dl, Pa, Qa, ba
    Cn = 32
    cand = torch.randint(0, I, (B, Cn), device=cuda, generator=gen, dtype=torch.int32)
    sel = torch.empty(B, device=cuda, dtype=torch.int32)
    for _ in range(3):
        dev.dns_select(P, Q, b, users, cand, out=sel)
    e0.record()
    for _ in range(steps):
        dev.dns_select(P, Q, b, users, cand, out=sel)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    per = (Cn + 1) * 4 * d + 8 * Cn
    gbs = per * B / (ms * 1e-3) / 1e9
    out["dns_select"] = {"positives_per_s": B / (ms * 1e-3), "ms": ms, "positives": B, "candidates": Cn,
                         "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_positive": per}}
    del cand, sel
    if not epoch_pipeline:
        return out
    # device epoch pipeline: draw one negative per stored interaction (rejection against the user's row),
    # shuffle, gather -- the work of model/BPR.py:50-62 + data/dataiterator.py:52-54 for one epoch
    deg = 16
    nnz = U * deg
    indptr = torch.arange(0, nnz + 1, deg, device=cuda, dtype=torch.int64)
    indices = torch.sort(torch.randint(0, I, (U, deg), device=cuda, generator=gen, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    su, sp, sn = (torch.empty(nnz, device=cuda, dtype=torch.int32) for _ in range(3))

    def epoch(e):
        dev.epoch_triples(indptr, indices, U, I, 2018, 2 * e, 2 * e + 1, users_out=su, pos_out=sp, neg_out=sn)
    epoch(0)
    torch.cuda.synchronize()
    e0.record()
    for e in range(1, 4):
        epoch(e)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out["device_epoch_pipeline"] = {"triples_per_s": nnz / (ms * 1e-3), "ms_per_epoch": ms, "interactions": nnz,
                                    "note": "tfrec_epoch_triples: sample + shuffle + gather of one epoch; the host sampler of the reference is O(items) per user on one core"}
    # one whole epoch of model/BPR.py:64-72 on the device: draw + shuffle + gather, then every step of the epoch
    # (batches of B triples, the last one short), on a versioned user table with the index of the next batch sorted ahead
    Pe, Qe, be_ = P.clone(), Q.clone(), b.clone()
    ut = dev.user_table(Pe)
    bounds = list(range(0, nnz, B)) + [nnz]

    def whole_epoch(e):
        epoch(e)
        dev.prepare_batch(0, U, I, su[bounds[0]:bounds[1]], sp[bounds[0]:bounds[1]], sn[bounds[0]:bounds[1]])
        for k in range(len(bounds) - 1):
            if k + 2 < len(bounds):
                lo, hi = bounds[k + 1], bounds[k + 2]
                dev.prepare_batch((k + 1) & 1, U, I, su[lo:hi], sp[lo:hi], sn[lo:hi])
            dev.bpr_sgd_step_prepared_v(k & 1, ut, Qe, be_, 0.01, 0.05)
        ut.flatten()
    whole_epoch(10)
    torch.cuda.synchronize()
    e0.record()
    for e in range(11, 14):
        whole_epoch(e)
    e1.record(); torch.cuda.synchronize()
    ut.close()
    ms = e0.elapsed_time(e1) / 3
    out["epoch_end_to_end"] = {"epoch_triples_per_s": nnz / (ms * 1e-3), "ms_per_epoch": ms, "interactions": nnz, "steps": len(bounds) - 1,
                               "batch": B, "note": "draw + shuffle + gather + every BPR step of the epoch + flatten, one stream, nothing overlapped "
                                                   "between epochs"}
    return out


def large_shape(dev, cuda, U=10_000_000, I=2_000_000, d=128, B=1 << 22, steps=10):
    """The training paths at the table shape of the largest configuration (10M users x 2M items, d=128; 6.1 GB of
    tables, far beyond the 256 MiB last-level cache), one GPU: BPR step on a versioned user table, APR step, DNS
    selection.  Timed outside the K steps of the headline metric."""
    g = torch.Generator(device=cuda)
    g.manual_seed(4242)
    P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.1
    Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.1
    b = torch.zeros(I, device=cuda)
    nb = 4
    users = torch.randint(0, U, (nb * B,), device=cuda, generator=g, dtype=torch.int32)
    pos = zipf_items(I, nb * B, g, cuda)
    neg = torch.randint(0, I, (nb * B,), device=cuda, generator=g, dtype=torch.int32)
    dev.reserve(B, d)
    ut = dev.user_table(P)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def grp(s):
        o = (s % nb) * B
        return users[o:o + B], pos[o:o + B], neg[o:o + B]

    def bpr(s):
        # (un-pipelined: at this shape sorting the next batch beside the row passes costs more than it hides)
        dev.bpr_sgd_step_v(ut, Q, b, *grp(s), 0.01, 0.05)
    for s in range(4):
        bpr(s)
    e0.record()
    for s in range(steps):
        bpr(4 + s)
    e1.record(); torch.cuda.synchronize()
    ut.close()
    ms = e0.elapsed_time(e1) / steps
    gbs = (24 * d + 28) * B / (ms * 1e-3) / 1e9
    res = {"shape": "%d users x %d items, d=%d, %d triples per step, one GPU" % (U, I, d, B),
           "bpr_sgd_step": {"triples_per_s": B / (ms * 1e-3), "ms_per_step": ms, "user_table": "versioned",
                            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                         "algorithmic_bytes_per_triple": 24 * d + 28,
                                         "traffic": pmc_traffic(B, d, "pmc_traffic_large.json") if (U, I) == (10_000_000, 2_000_000) else None,
                                         "traffic_source": "profiles/pmc_traffic_large.json (committed rocprofv3 counter passes of this shape), not measured in this run"}}}
    del users, pos, neg
    oth = other_paths(dev, P, Q, b, U, I, d, B, g, cuda, steps=steps, epoch_pipeline=False)
    res.update(oth)
    return res


def eval_block(dev, cuda, n, I, d, k, deg, tdeg, label):
    """Full-rank evaluation (model/BPR.py:76-80 + evaluator/evaluator.py:29-32) of n users against I items on
    freshly initialised tables (model/base/MatrixFactorization.py:17-21: factors ~ U(-0.05, 0.05)), so that no two
    users share a ranking.  Two bias settings: "independent" (b = 0, rankings unrelated between users: the worst
    case for the per-wave top-k filter) and "popularity_biased" (b ~ N(0, 3 x the spread of a factor dot product):
    the users of a wave prefer the same items, as after training).  users/s and the MFMA roofline of each."""
    from tfrec_amd.device import SCORE_F32, SCORE_BF16
    g = torch.Generator(device=cuda)
    g.manual_seed(777)
    P = (torch.rand(n, d, device=cuda, generator=g) - 0.5) * 0.1
    Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.1
    dot_std = (d ** 0.5) * (0.05 ** 2 / 3.0)
    eu = torch.arange(n, device=cuda, dtype=torch.int32)
    indptr = torch.arange(0, deg * (n + 1), deg, device=cuda, dtype=torch.int64)
    indices = torch.sort(torch.randint(0, I, (n, deg), device=cuda, generator=g, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    tptr = torch.arange(0, tdeg * (n + 1), tdeg, device=cuda, dtype=torch.int64)
    tidx = torch.sort(torch.randint(0, I, (n, tdeg), device=cuda, generator=g, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    res = {"shape": "%s: %d users x %d items, d=%d, k=%d, %d train + %d test items per user" % (label, n, I, d, k, deg, tdeg)}
    for case, sigma in (("independent", 0.0), ("popularity_biased", 3.0 * dot_std)):
        b = torch.randn(I, device=cuda, generator=g) * sigma
        ev = {}
        for mode, name in ((SCORE_F32, "f32"), (SCORE_BF16, "bf16")):
            dev.eval_fullrank(P, Q, b, eu, (indptr, indices), (tptr, tidx), k, mode=mode)      # warm-up, sizes the workspace
            torch.cuda.synchronize()
            reps = 3 if name == "bf16" else 1
            t1 = time.perf_counter()
            for _ in range(reps):
                dev.eval_fullrank(P, Q, b, eu, (indptr, indices), (tptr, tidx), k, mode=mode)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            tf = 2.0 * I * d * n / dt / 1e12
            ev[name] = {"users_per_s": n / dt, "ms": dt * 1e3,
                        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS[name], "unit": "TFLOP/s",
                                     "frac": tf / MFMA_PEAK_TFLOPS[name], "traffic": None}}
            if name == "bf16" and SUSTAINED_BF16_TFLOPS:
                # beside the nominal peak: what the instruction sustains on random operands at the clock the chip then holds
                ev[name]["roofline"]["sustained_peak"] = SUSTAINED_BF16_TFLOPS
                ev[name]["roofline"]["frac_of_sustained"] = tf / SUSTAINED_BF16_TFLOPS
        res[case] = ev
    return res


def cpu_baseline(U, I, d, B, lr, reg, budget_s=12.0):
    """The CPU oracle (plain C, OpenMP over row owners) on the same table shapes and batch distribution, on the
    host cores this process may use (the box sets OMP_NUM_THREADS to its share)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    rng = np.random.default_rng(2018)
    P = rng.uniform(-0.05, 0.05, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.05, 0.05, (I, d)).astype(np.float32)
    b = np.zeros(I, np.float32)
    w = 1.0 / np.arange(1, I + 1)
    cdf = np.cumsum(w) / w.sum()
    perm = rng.permutation(I)
    ws = orc.Workspace(U, I, d)
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1, 64))

    def draw():
        return (rng.integers(0, U, B).astype(np.int32), perm[np.searchsorted(cdf, rng.random(B)).clip(max=I - 1)].astype(np.int32),
                rng.integers(0, I, B).astype(np.int32))
    for _ in range(2):          # untimed: first touch of the accumulators, thread pool start
        orc.bpr_step(P, Q, b, *draw(), lr, reg, ws=ws, threads=cores)
    done, t_used, steps = 0, 0.0, 0
    while t_used < budget_s:
        users, pos, neg = draw()
        t0 = time.perf_counter()
        orc.bpr_step(P, Q, b, users, pos, neg, lr, reg, ws=ws, threads=cores)
        t_used += time.perf_counter() - t0
        done += B
        steps += 1
    one, t1 = 0, 0.0
    while t1 < 3.0:
        users, pos, neg = draw()
        t0 = time.perf_counter()
        orc.bpr_step(P, Q, b, users, pos, neg, lr, reg, ws=ws)
        t1 += time.perf_counter() - t0
        one += B
    return {"value": done / t_used, "unit": "triples/s", "cores": cores, "kind": "port", "one_core_value": one / t1,
            "sample": "%d steps of %d triples on the same %dx%d d=%d tables (oracle/mf_oracle.c, C with OpenMP over row owners, "
                      "step math only; no TensorFlow number exists)" % (steps, B, U, I, d)}


def cpu_baseline_eval(I, d, k, n_users=1024, threads=16):
    """Full-rank evaluation of a bounded user sample on the host: the reference's own C++ top-k and metrics
    (compiled from its sources into oracle/_ref, evaluator/backend/cpp/tools.h:10-30, evaluate.h:13-93) behind
    the dense numpy scoring of model/BPR.py:76-80, when that library is present; else the single-thread C
    restatement."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    rng = np.random.default_rng(7)
    P = rng.uniform(-0.05, 0.05, (n_users, d)).astype(np.float32)
    Q = rng.uniform(-0.05, 0.05, (I, d)).astype(np.float32)
    b = rng.normal(size=I).astype(np.float32)
    deg, tdeg = 20, 3
    train = np.sort(rng.integers(0, I, (n_users, deg)), axis=1).astype(np.int32)
    test = [np.unique(rng.integers(0, I, tdeg)).astype(np.int32) for _ in range(n_users)]
    if orc.ref_lib() is not None:
        t0 = time.perf_counter()
        R = np.matmul(P, Q.T) + b
        np.put_along_axis(R, train.astype(np.int64), -np.inf, axis=1)
        ranks = orc.ref_topk_rows(R, k, threads=threads)
        orc.ref_evaluate(ranks, test, threads=threads)
        dt = time.perf_counter() - t0
        return {"value": n_users / dt, "unit": "users/s", "cores": threads, "kind": "reference",
                "sample": "%d users x %d items, d=%d, k=%d: numpy scoring + the reference's C++ top-k and metrics (oracle/_ref), %d threads"
                          % (n_users, I, d, k, threads)}
    n_users = min(n_users, 128)
    tptr = np.arange(0, deg * (n_users + 1), deg, dtype=np.int64)
    eptr = np.zeros(n_users + 1, np.int64)
    eptr[1:] = np.cumsum([len(t) for t in test[:n_users]])
    t0 = time.perf_counter()
    orc.eval_fullrank(P[:n_users], Q, b, np.arange(n_users, dtype=np.int32), (tptr, train[:n_users].ravel()),
                      (eptr, np.concatenate(test[:n_users])), k, score_mode=0)
    dt = time.perf_counter() - t0
    return {"value": n_users / dt, "unit": "users/s", "cores": 1, "kind": "port",
            "sample": "%d users x %d items, d=%d, k=%d (oracle/mf_oracle.c, scalar C)" % (n_users, I, d, k)}


def large_shape_sharded(dev, cuda, world, rank, scale, barrier, max_over_ranks, steps=8):
    """The largest configuration's layout over the ranks (weak scaling: per GPU 1.25M users x 250k items of d=128 -- 10M x 2M
    on eight GPUs -- and 4M triples per step): every table sharded, planned all-to-all steps (with the split of a step
    into fetch / compute / return / apply from stream events), the DNS draw of an epoch with 32 candidates per positive
    (candidate pools, one all-gather of the item shards, selection), and the item-sharded bf16 top-100 evaluation."""
    from tfrec_amd.device import SCORE_BF16
    from tfrec_amd.sharded import PhaseTimers, ShardedBprTrainer, ShardedEpochSampler, ShardedEvaluator
    d = 128
    nu, ni, B = max(int(1_250_000 * scale), 1024), max(int(250_000 * scale), 512), max(int((1 << 22) * scale), 4096)
    U, I = nu * world, ni * world
    tr = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=4242, device=cuda, item_mode="sharded")
    lo, hi = tr.user_range
    g = torch.Generator(device=cuda)
    g.manual_seed(4242 + rank)
    nb = 4
    users = torch.randint(0, hi - lo, (nb * B,), device=cuda, generator=g, dtype=torch.int32) + lo
    pos = zipf_items(I, nb * B, g, cuda)
    neg = torch.randint(0, I, (nb * B,), device=cuda, generator=g, dtype=torch.int32)
    batches = [(users[o:o + B], pos[o:o + B], neg[o:o + B]) for o in range(0, nb * B, B)]
    res = {"shape": "%d GPUs x (%d users, %d items), d=%d, %d triples per step and GPU; every table sharded by rows" % (world, nu, ni, d, B)}

    def timed(fn, reps):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return max_over_ranks((time.perf_counter() - t0) / reps)

    def epoch_of_steps():
        tr.run_planned(tr.plan_steps(batches))
    dt = timed(epoch_of_steps, max(steps // nb, 1)) / nb
    per_gpu = B * (24 * d + 28) / dt / 1e9
    tr._timers = PhaseTimers()
    epoch_of_steps()
    split = {k: v / nb for k, v in tr._timers.totals().items()}
    tr._timers = None
    res["bpr_sgd_step"] = {"triples_per_s": world * B / dt, "ms_per_step": dt * 1e3, "split_ms": split,
                           "note": "plan (de-duplication, counts and ids exchanged once for the %d steps) + steps; split: fetch = owners' rows "
                                   "gathered and exchanged, compute = local step, return = gradient sums exchanged, apply = owners' update" % nb,
                           "roofline": {"bound": "hbm", "achieved": per_gpu, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU", "frac": per_gpu / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_triple": 24 * d + 28}}
    del users, pos, neg, batches
    # DNS: the epoch's candidate pools (32 per positive), the item shards gathered once, the best-scored candidate
    deg, Cn = 10, 32
    indptr = torch.arange(0, deg * (nu + 1), deg, device=cuda, dtype=torch.int64)
    indices = torch.sort(torch.randint(0, I, (nu, deg), device=cuda, generator=g, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    sm = ShardedEpochSampler.from_local_csr(dev, tr, indptr, indices, first=rank * nu * deg, total_nnz=U * deg, seed=2018, n_neg=Cn)
    state = {"e": 0}

    def draw():
        state["e"] += 1
        sm.draw(state["e"])
    dt = timed(draw, 2)
    per = (Cn + 1) * 4 * d + 8 * Cn
    gbs = per * nu * deg / dt / 1e9
    res["dns_epoch_draw"] = {"positives_per_s": world * nu * deg / dt, "ms": dt * 1e3, "positives_per_gpu": nu * deg, "candidates": Cn,
                             "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU", "frac": gbs / HBM_PEAK_GBS,
                                          "algorithmic_bytes_per_positive": per}}
    del sm, indptr, indices
    # evaluation: bf16 scores, top 100, item-sharded; 131072 evaluated users per GPU
    g2 = torch.Generator(device=cuda)
    g2.manual_seed(777)                                      # the same lists on every rank
    n_eval, k, tdeg = min(max(int(131072 * scale), 256) * world, U), 100, 3
    eu = (torch.arange(n_eval, device=cuda, dtype=torch.int64) * U // n_eval).to(torch.int32)
    tptr = torch.arange(0, deg * (U + 1), deg, device=cuda, dtype=torch.int64)
    tidx = torch.sort(torch.randint(0, I, (U, deg), device=cuda, generator=g2, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    eptr = torch.arange(0, tdeg * (U + 1), tdeg, device=cuda, dtype=torch.int64)
    eidx = torch.sort(torch.randint(0, I, (U, tdeg), device=cuda, generator=g2, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    fresh = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=777, device=cuda, item_mode="sharded")
    se = ShardedEvaluator(dev, fresh, (tptr, tidx), (eptr, eidx), eu, top_k=k, mode=SCORE_BF16, chunk_users=max(n_eval // world, 1))
    dt = timed(se.evaluate, 1)
    tf = 2.0 * I * d * n_eval / dt / 1e12
    res["eval_bf16_top100"] = {"users_per_s": n_eval / dt, "ms": dt * 1e3, "users": n_eval, "items": I, "layout": se.layout + "-sharded",
                               "roofline": {"bound": "mfma", "achieved": tf / world, "peak": MFMA_PEAK_TFLOPS["bf16"], "unit": "TFLOP/s per GPU",
                                            "frac": tf / world / MFMA_PEAK_TFLOPS["bf16"]}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1 << 22, help="triples per step per GPU")
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=200_000)
    ap.add_argument("--factors", type=int, default=64)
    ap.add_argument("--interactions", type=int, default=100_000_000)
    ap.add_argument("--eval-users", type=int, default=262_144)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the evaluation block at the 2M-item, d=128, k=100 shape")
    ap.add_argument("--no-large", action="store_true", help="skip the BPR / APR / DNS timings at the 10M-user x 2M-item, d=128 shape")
    ap.add_argument("--no-other", action="store_true", help="skip the APR step and DNS selection timings")
    ap.add_argument("--prep-batches", type=int, default=1, help="batches sorted per index preparation (single GPU)")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU code path even on one GPU (testing)")
    ap.add_argument("--item-mode", default="auto", choices=["auto", "dense", "sharded"],
                    help="multi-GPU item table: replicated with one all-reduce of gradient sums per step, or sharded with all-to-all lookups")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend for N>1 (gloo: testing only)")
    ap.add_argument("--overlap-chunks", type=int, default=None, help="replicated item table: ranges of item ids whose all-reduce "
                    "overlaps the rest of the item half (default: 2 under RCCL with more than one rank, else 1)")
    ap.add_argument("--force-dist", action="store_true", help="testing: with one rank, still create the process group, so that "
                    "every collective of the multi-GPU path runs (trivially) through the chosen backend")
    ap.add_argument("--single-device", action="store_true", help="testing: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--plain-table", action="store_true", help="single GPU: plain user table (pre-step rows staged per triple) instead of the versioned one")
    ap.add_argument("--no-extras", action="store_true", help="skip the timing of the reference batch size (512 triples per step)")
    ap.add_argument("--secondary-timeout", type=int, default=480,
                    help="N > 1: seconds after the headline measurement at which the line is printed without the blocks still running")
    ap.add_argument("--large-scale", type=float, default=1.0, help="testing: shrink the sharded large-shape block (N > 1) by this factor")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # before anything initialises the GPU in this process; the ranks are fresh child processes, all of them watched
        from tfrec_amd.launch import launch_ranks
        raise SystemExit(launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    cuda = torch.device("cuda", local_rank)
    if world > 1 or args.force_dist:
        from tfrec_amd.launch import init_process_group
        init_process_group(args.backend, device=cuda)

    from tfrec_amd.device import Device, SCORE_F32, SCORE_BF16
    stream = torch.cuda.Stream(device=cuda)
    lr, reg = 0.01, 0.05                                   # conf/BPR.ini
    U, I, d, B = args.users, args.items, args.factors, args.batch
    gen = torch.Generator(device=cuda)
    gen.manual_seed(2018 + rank)
    n_batches = max(1, min(args.interactions // (B * world), 16))    # distinct batches resident in HBM, cycled

    with torch.cuda.stream(stream):
        dev = Device(local_rank, stream=stream.cuda_stream)
        if world == 1 and not args.force_sharded:
            P = (torch.rand(U, d, device=cuda, generator=gen) - 0.5) * 0.1     # U(-0.05, 0.05)
            Q = (torch.rand(I, d, device=cuda, generator=gen) - 0.5) * 0.1
            b = torch.zeros(I, device=cuda)
            users = torch.randint(0, U, (n_batches * B,), device=cuda, generator=gen, dtype=torch.int32)
            pos = zipf_items(I, n_batches * B, gen, cuda)
            neg = torch.randint(0, I, (n_batches * B,), device=cuda, generator=gen, dtype=torch.int32)
            dev.reserve(B, d)
            # the user table is versioned (tfrec_utable): steps need not stage the pre-step user rows for their
            # item half; P itself is complete again after ut.flatten()
            ut = dev.user_table(P) if not args.plain_table else None

            def batch(s):
                o = (s % n_batches) * B
                return users[o:o + B], pos[o:o + B], neg[o:o + B]

            # the index work of batch s+1 is enqueued on the library's auxiliary stream before the row
            # passes of batch s, so the two overlap; every step still does all of its own work
            # (K batches are sorted per preparation, so the fixed costs of the index phase are paid once per K steps)
            K = args.prep_batches
            assert n_batches % K == 0

            def group(g):
                o = ((g * K) % n_batches) * B
                return users[o:o + K * B], pos[o:o + K * B], neg[o:o + K * B]

            state = {"next": 0}
            dev.prepare_batches(0, U, I, *group(0), K)

            def step(s):
                k = state["next"]
                g, r = divmod(k, K)
                if r == 0:
                    dev.prepare_batches((g + 1) & 1, U, I, *group(g + 1), K)
                if ut is not None:
                    dev.bpr_sgd_step_prepared_v(g & 1, ut, Q, b, lr, reg)
                else:
                    dev.bpr_sgd_step_prepared(g & 1, P, Q, b, lr, reg)
                state["next"] = k + 1
            trainer = None
        else:
            from tfrec_amd.sharded import ShardedBprTrainer
            trainer = ShardedBprTrainer(dev, U, I, d, lr, reg, seed=2018, device=cuda, item_mode=args.item_mode,
                                        overlap_chunks=args.overlap_chunks)
            lo, hi = trainer.user_range
            users = (torch.randint(0, hi - lo, (n_batches * B,), device=cuda, generator=gen, dtype=torch.int32) + lo)
            pos = zipf_items(I, n_batches * B, gen, cuda)
            neg = torch.randint(0, I, (n_batches * B,), device=cuda, generator=gen, dtype=torch.int32)

            def batch(s):
                o = (s % n_batches) * B
                return users[o:o + B], pos[o:o + B], neg[o:o + B]

            if trainer.item_mode == "dense":
                # replicated item table: the index work of the next batch runs on the auxiliary stream
                # beside this step's all-reduce and update
                state = {"next": 0}
                trainer.prepare(0, *batch(0))

                def step(s):
                    k = state["next"]
                    trainer.step_prepared(k & 1, prepare_next=((k + 1) & 1,) + tuple(batch(k + 1)))
                    state["next"] = k + 1
            else:
                # every table sharded: the lookups of the next n_batches steps are planned together (de-duplication
                # on the device, one exchange of all the counts, one host read), as an epoch loop does
                plans = []

                def step(s):
                    if not plans:
                        plans.extend(trainer.plan_steps([batch(s + i) for i in range(n_batches)]))
                    plan = plans.pop(0)
                    trainer.step_planned(plan, slot=s & 1, next_plan=plans[0] if plans else None)

        def max_over_ranks(x):
            if not (world > 1 or args.force_dist):
                return x
            # (a host tensor under gloo: torch's gloo backend copies device tensors on high-priority streams, which starves
            # the other rank when ranks share a GPU)
            t = torch.tensor([x], device=cuda if args.backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())

        def barrier():
            if world > 1 or args.force_dist:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        for s in range(args.warmup):
            step(s)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for s in range(args.steps):
            step(args.warmup + s)
        e1.record(stream)
        barrier()
        wall = time.perf_counter() - t0
        dev_ms = e0.elapsed_time(e1)
        if world > 1 or args.force_dist:
            wall = max_over_ranks(wall)
        dev.synchronize()        # raises if any batch held an out-of-range index

        ms_per_step = wall * 1e3 / args.steps
        value = world * B * args.steps / wall
        alg_bytes_per_triple = 24 * d + 28
        out = {
            "metric": "bpr_train_triples_per_s", "value": value, "unit": "triples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BPR-MF SGD step, synthetic %dM users x %dk items, d=%d fp32, %d-triple batch per GPU, "
                                   "users uniform, positives Zipf(1.0), negatives uniform" % (U // 10**6, I // 1000, d, B),
                       "users": U, "items": I, "factors": d, "batch_per_gpu": B, "global_batch": B * world,
                       "lr": lr, "reg": reg, "resident_batches": n_batches,
                       "parallelism": "single GPU" if trainer is None else (
                           "user table row-sharded over %d GPUs, item table replicated, all-reduce of item gradient sums per step in %d range(s) of item ids"
                           % (world, trainer.overlap_chunks)
                           if trainer.item_mode == "dense" else "tables row-sharded over %d GPUs, all-to-all lookups per step" % world)},
        }

        if trainer is not None:
            # multi-GPU path: per-GPU algorithmic rate over the whole step (exchange included)
            per_gpu = B * alg_bytes_per_triple / (wall / args.steps) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": per_gpu, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per_gpu / HBM_PEAK_GBS,
                               "traffic": None, "kernel": "per GPU: one BPR step incl. the item-gradient exchange",
                               "algorithmic_bytes_per_triple": alg_bytes_per_triple}
        # Several ranks: the secondary measurements below are collective.  If a rank fails inside one of them (the others then
        # wait in a collective for ever) or a block outgrows its time, the headline line is not held back: after
        # --secondary-timeout seconds rank 0 prints what is complete and every rank leaves.
        emitted = threading.Lock()

        def emit(note=None):
            if not emitted.acquire(blocking=False):
                return
            if rank == 0:
                line = dict(out)
                if note:
                    line["secondary_timeout"] = note
                print(json.dumps(line), flush=True)

        watchdog = None
        if world > 1 or args.force_dist:
            def give_up():
                emit("secondary measurements stopped after %d s; the blocks present are complete" % args.secondary_timeout)
                sys.stdout.flush()
                os._exit(0)
            watchdog = threading.Timer(args.secondary_timeout + (0 if rank == 0 else 5), give_up)
            watchdog.daemon = True
            watchdog.start()

        def guarded(name, fn):
            """Secondary measurements must not cost the headline line: a failure is reported in their place."""
            try:
                out[name] = fn()
            except Exception as e:      # noqa: BLE001 -- reported, not swallowed
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

        if trainer is not None and not args.no_eval:
            def sharded_eval():
                # full-rank evaluation over the ranks, eval_users per GPU (weak scaling).  Replicated item table:
                # every rank evaluates the users it owns against all items.  Sharded item table: every rank scores
                # its item range for all evaluated users (factors broadcast from the owner), lists gathered and merged.
                from tfrec_amd.sharded import ShardedEvaluator
                g2 = torch.Generator(device=cuda)
                g2.manual_seed(4242)                                   # the same interaction lists on every rank
                n, k = min(args.eval_users * world, U), args.topk
                eu = (torch.arange(n, device=cuda, dtype=torch.int64) * U // n).to(torch.int32)     # spread evenly over the owners
                deg, tdeg = 20, 3
                indptr = torch.arange(0, deg * (U + 1), deg, device=cuda, dtype=torch.int64)
                indices = torch.sort(torch.randint(0, I, (U, deg), device=cuda, generator=g2, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
                tptr = torch.arange(0, tdeg * (U + 1), tdeg, device=cuda, dtype=torch.int64)
                tidx = torch.sort(torch.randint(0, I, (U, tdeg), device=cuda, generator=g2, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
                # freshly initialised tables of the same layout (the trained ones rank every user alike)
                fresh = ShardedBprTrainer(dev, U, I, d, lr, reg, seed=4242, device=cuda, item_mode=trainer.item_mode)
                ev = {"model": "fresh tables, factors ~ U(-0.05, 0.05), b = 0 (independent rankings)"}
                for mode, name in ((SCORE_F32, "f32"), (SCORE_BF16, "bf16")):
                    se = ShardedEvaluator(dev, fresh, (indptr, indices), (tptr, tidx), eu, top_k=k, mode=mode)
                    se.evaluate()                                      # warm-up, sizes the workspaces
                    barrier()
                    t1 = time.perf_counter()
                    se.evaluate()
                    barrier()
                    dt = time.perf_counter() - t1
                    if world > 1 or args.force_dist:
                        dt = max_over_ranks(dt)
                    tf = 2.0 * I * d * n / dt / 1e12
                    ev[name] = {"users_per_s": n / dt, "ms": dt * 1e3, "users": n, "items": I, "topk": k, "layout": se.layout + "-sharded",
                                "roofline": {"bound": "mfma", "achieved": tf / world, "peak": MFMA_PEAK_TFLOPS[name], "unit": "TFLOP/s per GPU",
                                             "frac": tf / world / MFMA_PEAK_TFLOPS[name]}}
                return ev
            guarded("eval", sharded_eval)
        if trainer is not None and not args.no_other:
            def sharded_apr():
                # the APR step over the ranks (conf/APR.ini): user tables sharded, item tables replicated, two
                # all-reduces of item-gradient sums per step
                from tfrec_amd.sharded import ShardedAprTrainer
                apr = ShardedAprTrainer(dev, U, I, d, 0.05, 0.0, 1.0, 0.05, seed=2018, device=cuda)
                for s in range(3):
                    apr.step(*batch(s), global_batch=B * world)
                barrier()
                t1 = time.perf_counter()
                apr_steps = 10
                for s in range(apr_steps):
                    apr.step(*batch(3 + s), global_batch=B * world)
                barrier()
                dt = time.perf_counter() - t1
                if world > 1 or args.force_dist:
                    dt = max_over_ranks(dt)
                apr_bytes = 84 * d + 68
                per_gpu = B * apr_bytes / (dt / apr_steps) / 1e9
                res = {"apr_adagrad_step": {
                    "triples_per_s": world * B * apr_steps / dt, "ms_per_step": dt * 1e3 / apr_steps, "batch_per_gpu": B,
                    "parallelism": "user tables row-sharded over %d GPUs, item tables replicated, two all-reduces per step" % world,
                    "roofline": {"bound": "hbm", "achieved": per_gpu, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU", "frac": per_gpu / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_triple": apr_bytes}}}
                del apr
                return res
            guarded("other_paths", sharded_apr)
        if trainer is not None and not args.no_large:
            torch.cuda.empty_cache()
            guarded("large_shape", lambda: large_shape_sharded(dev, cuda, world, rank, args.large_scale, barrier, max_over_ranks))
        if rank == 0 and world == 1 and not args.force_sharded:
            # ---- roofline of the step's kernels: HIP events recorded by the library on its own stream
            # (un-pipelined steps, so that the index phase is timed on the same stream as the row passes)
            dev.set_profiling(True)
            acc, reps = {}, 10
            for s in range(reps):
                if ut is not None:
                    dev.bpr_sgd_step_v(ut, Q, b, *batch(s), lr, reg)
                else:
                    dev.bpr_sgd_step(P, Q, b, *batch(s), lr, reg)
                for name, t in dev.phase_times():
                    acc[name] = acc.get(name, 0.0) + t / reps
            dev.set_profiling(False)
            if ut is not None:
                ut.flatten()            # P is complete again for what follows
            step_ms = sum(acc.values())
            dom = max(acc, key=acc.get)
            achieved = B * alg_bytes_per_triple / (dev_ms / args.steps * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(B, d),
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                                  "committed with the profiles), not measured in this run",
                "kernel": "one BPR step = radix sort of the (row, occurrence) pairs, reading the triples in its first pass + seg_tile/seg_fixup (user side) + seg_tile/seg_fixup (item side)"
                          + (" + version flip, on a versioned user table" if ut is not None else ""),
                "algorithmic_bytes_per_triple": alg_bytes_per_triple,
                # the streaming-copy rate this chip reaches in practice (the microarchitecture guide's measured figure)
                "measured_copy_peak": MEASURED_COPY_GBS, "frac_of_measured_copy_peak": achieved / MEASURED_COPY_GBS,
                "device_ms_per_step": dev_ms / args.steps,
                "phases_ms": acc, "dominant_phase": dom,
            }
            if not args.no_extras:
                # positives uniform over the items instead of Zipf: the bound without hot rows (fewer duplicates,
                # fewer cache hits on item rows)
                e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                upos = torch.randint(0, I, (4 * B,), device=cuda, generator=gen, dtype=torch.int32)
                ut_u = dev.user_table(P) if ut is not None else None

                def ustep(s):
                    o = (s % 4) * B
                    if ut_u is not None:
                        dev.bpr_sgd_step_v(ut_u, Q, b, users[o:o + B], upos[o:o + B], neg[o:o + B], lr, reg)
                    else:
                        dev.bpr_sgd_step(P, Q, b, users[o:o + B], upos[o:o + B], neg[o:o + B], lr, reg)
                for s in range(4):
                    ustep(s)
                e2.record(stream)
                for s in range(20):
                    ustep(s)
                e3.record(stream)
                torch.cuda.synchronize()
                if ut_u is not None:
                    ut_u.close()
                ums = e2.elapsed_time(e3) / 20
                out["uniform_items_step"] = {"ms_per_step": ums, "triples_per_s": B / (ums * 1e-3),
                                             "frac": B * alg_bytes_per_triple / (ums * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "note": "un-pipelined step, positives uniform over the items (no hot rows)"}
                del upos
                # batches of 2^20 triples on the same tables: about one occurrence per user, so no user row is shared
                # between the triples of a batch (the prepared, pipelined path of the headline)
                Bs = 1 << 20
                if B > Bs and B % Bs == 0:
                    nsm = (n_batches * B) // Bs

                    def sm(k):
                        o = (k % nsm) * Bs
                        return users[o:o + Bs], pos[o:o + Bs], neg[o:o + Bs]
                    ut_s = dev.user_table(P) if ut is not None else None
                    dev.prepare_batches(0, U, I, *sm(0), 1)
                    for k in range(64 + 8):
                        if k == 8:
                            e2.record(stream)
                        dev.prepare_batches((k + 1) & 1, U, I, *sm(k + 1), 1)
                        if ut_s is not None:
                            dev.bpr_sgd_step_prepared_v(k & 1, ut_s, Q, b, lr, reg)
                        else:
                            dev.bpr_sgd_step_prepared(k & 1, P, Q, b, lr, reg)
                    e3.record(stream)
                    torch.cuda.synchronize()
                    if ut_s is not None:
                        ut_s.close()
                    sms = e2.elapsed_time(e3) / 64
                    out["batch_2p20"] = {"ms_per_step": sms, "triples_per_s": Bs / (sms * 1e-3), "batch": Bs,
                                         "frac": Bs * alg_bytes_per_triple / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "note": "same tables and path as the headline, 2^20 triples per step (about one occurrence per user)"}
                # the reference batch size (conf/BPR.ini:4: 512 triples per step): five short kernels, latency-bound
                small = 512
                ref = {}
                for name, fn in (("bpr_sgd_step", dev.bpr_sgd_step), ("bpr_sgd_step_noalloc", dev.bpr_sgd_step_noalloc)):
                    for s in range(20):
                        o = s * small
                        fn(P, Q, b, users[o:o + small], pos[o:o + small], neg[o:o + small], lr, reg)
                    e2.record(stream)
                    for s in range(200):
                        o = (s % 1000) * small
                        fn(P, Q, b, users[o:o + small], pos[o:o + small], neg[o:o + small], lr, reg)
                    e3.record(stream)
                    torch.cuda.synchronize()
                    ref[name] = {"triples_per_s": 200 * small / (e2.elapsed_time(e3) * 1e-3), "us_per_step": e2.elapsed_time(e3) * 1e3 / 200}
                out["reference_batch_512"] = ref

            if not args.no_other and not args.no_large:
                # (before the evaluation blocks and on memory fresh from the driver: the 5 GB tables of this block run
                # about 8 % slower on blocks the caching allocator has recycled from the evaluation's tensors)
                torch.cuda.empty_cache()
                guarded("large_shape", lambda: large_shape(dev, cuda))
                torch.cuda.empty_cache()
            if not args.no_eval:
                # the trained tables are not evaluated: 400 summed-gradient steps at this batch size leave item
                # biases that make every user's ranking the same, the easiest case for the top-k filter
                ev = eval_block(dev, cuda, min(args.eval_users, U), I, d, args.topk, 20, 3, "C2 shape")
                if not args.no_c5:
                    c5 = eval_block(dev, cuda, args.eval_users, 2_000_000, 128, 100, 10, 3, "C5 shape")
                    ev["c5"] = c5
                    ev["roofline"] = dict(c5["independent"]["bf16"]["roofline"],
                                          kernel="score_topk_pp_kernel (bf16 MFMA 32x32x16) + metrics, whole call",
                                          workload="%d users x 2M items, d=128, k=100, independent factors" % args.eval_users)
                out["eval"] = ev
            if not args.no_other:
                out["other_paths"] = other_paths(dev, P, Q, b, U, I, d, B, gen, cuda)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(U, I, d, 1 << 17, lr, reg)
                if "eval" in out and "error" not in out["eval"]:
                    out["eval"]["cpu_baseline"] = cpu_baseline_eval(I, d, args.topk)
        if watchdog is not None:
            watchdog.cancel()
        emit()
    if world > 1 or args.force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
