"""Developer probe: the device epoch pipeline alone (draw + shuffle + gather of 16M interactions, 1M users x 200k items),
for a kernel trace.  python tools/epoch_probe.py [epochs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device  # noqa: E402

cuda = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
dev = Device(0, stream=stream.cuda_stream)
U, I, deg = 1_000_000, 200_000, 16
nnz = U * deg
gen = torch.Generator(device=cuda); gen.manual_seed(1)
indptr = torch.arange(0, nnz + 1, deg, device=cuda, dtype=torch.int64)
indices = torch.sort(torch.randint(0, I, (U, deg), device=cuda, generator=gen, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
eu, ep, en = (torch.empty(nnz, device=cuda, dtype=torch.int32) for _ in range(3))
su, sp, sn = (torch.empty(nnz, device=cuda, dtype=torch.int32) for _ in range(3))
perm = torch.empty(nnz, device=cuda, dtype=torch.int32)


def epoch(e):
    if os.environ.get("EPOCH_PROBE_THREE_CALLS"):
        dev.sample_negatives(indptr, indices, U, I, 1, 2018, 2 * e, users_out=eu, pos_out=ep, neg_out=en)
        dev.shuffle_perm(nnz, 2018, 2 * e + 1, out=perm)
        dev.gather_rows_i32(eu, perm, out=su); dev.gather_rows_i32(ep, perm, out=sp); dev.gather_rows_i32(en, perm, out=sn)
    else:
        dev.epoch_triples(indptr, indices, U, I, 2018, 2 * e, 2 * e + 1, users_out=su, pos_out=sp, neg_out=sn)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
epoch(0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for e in range(1, 1 + n):
    epoch(e)
e1.record(); torch.cuda.synchronize()
print("epoch pipeline: %.3f ms per epoch of %d interactions" % (e0.elapsed_time(e1) / n, nnz))
