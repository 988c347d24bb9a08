"""Developer tool: per-wave timeline of 64 tiles of one workgroup of the second scoring kernel (a -DTF_EVAL_PROFILE build
named by TFREC_AMD_LIB): when each wave started a tile, finished its blocks, entered and left the barrier."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tfrec_amd.device import Device, SCORE_BF16


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "biased"
    ablate_note = sys.argv[2] if len(sys.argv) > 2 else ""
    cuda = torch.device("cuda", 0)
    dev = Device(0, stream=torch.cuda.current_stream().cuda_stream)
    n, I, d, k = 262144, 2_000_000, 128, 100
    g = torch.Generator(device=cuda); g.manual_seed(777)
    P = (torch.rand(n, d, device=cuda, generator=g) - 0.5) * 0.1
    Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.1
    sigma = 0.0 if model == "random" else 3.0 * (d ** 0.5) * (0.05 ** 2 / 3.0)
    b = torch.randn(I, device=cuda, generator=g) * sigma
    eu = torch.arange(n, device=cuda, dtype=torch.int32)
    ids = torch.empty(n, k, device=cuda, dtype=torch.int32); sc = torch.empty(n, k, device=cuda)
    dev.score_topk(P, Q, b, eu, k, mode=SCORE_BF16, out_ids=ids, out_scores=sc)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 64 * 4))()
    assert dev.lib.tfrec_debug_trace(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(8, 64, 4)
    t0 = t[:, 0, 0].min()
    print("model %s %s: cycles relative to the first tile start; per tile: wave: start / blocks done / barrier in / barrier out" % (model, ablate_note))
    for tile in range(0, 24):
        print("tile %2d: " % tile + "  ".join("w%d %6d/%6d/%6d/%6d" % (w, *(t[w, tile] - t0)) for w in range(8)))
    dur = t[:, 1:, 0] - t[:, :-1, 0]
    print("tile period per wave (mean over 63 tiles):", dur.mean(axis=1).round(0))
    print("blocks phase (mean):", (t[:, :, 1] - t[:, :, 0]).mean(axis=1).round(0))
    print("end-of-tile work (mean):", (t[:, :, 2] - t[:, :, 1]).mean(axis=1).round(0))
    print("barrier wait (mean):", (t[:, :, 3] - t[:, :, 2]).mean(axis=1).round(0))


if __name__ == "__main__":
    main()
