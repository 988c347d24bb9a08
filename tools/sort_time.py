"""Developer probe: time the pair sort for given sizes and key widths.  python tools/sort_time.py n:bits ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
dev = Device(0, stream=stream.cuda_stream)
for spec in sys.argv[1:]:
    n, bits = (int(x) for x in spec.split(":"))
    keys = torch.randint(0, 1 << bits, (n,), device="cuda", dtype=torch.int64).to(torch.int32)
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    ok, ov = torch.empty_like(keys), torch.empty_like(vals)
    for _ in range(3):
        dev.sort_pairs(keys, vals, bits, out_keys=ok, out_vals=ov)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dev.sort_pairs(keys, vals, bits, out_keys=ok, out_vals=ov)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"n={n} bits={bits}: {ms*1e3:.1f} us  {n/ms/1e6:.2f} G pairs/s", flush=True)
