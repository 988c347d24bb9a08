"""Developer probe: what the scoring kernels return for NaN / inf scores."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device, SCORE_F32, SCORE_BF16
dev = Device(0)
rng = np.random.default_rng(3)
U, I, d, k = 70, 900, 32, 20
P = rng.normal(size=(U, d)).astype(np.float32)
Q = rng.normal(size=(I, d)).astype(np.float32)
b = np.zeros(I, np.float32)
b[5] = np.inf; b[17] = np.inf; b[33] = -np.inf; b[40] = np.nan; b[41] = np.nan
Q[100, 3] = np.nan
P[7, :] = np.nan            # a diverged user: every score NaN
eu = np.arange(U, dtype=np.int32)
for mode in (SCORE_F32, SCORE_BF16):
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(eu, np.int32), k, mode=mode)
    dev.synchronize()
    ids, sc = ids.numpy(), sc.numpy()
    print("mode", mode, "user0", ids[0][:8], sc[0][:8])
    print("  user7 (NaN row)", ids[7][:8], sc[7][:8])
    bad = [u for u in range(U) if u != 7 and (set(ids[u]) & {40, 41, 100})]
    print("  users with a NaN-scored item in the list:", bad[:5], " id range", ids.min(), ids.max(),
          " dup rows", sum(len(set(r)) != k for r in ids))
R = (P @ Q.T + b).astype(np.float32)
ids, sc = dev.topk_dense(dev.array(R), k)
dev.synchronize()
ids, sc = ids.numpy(), sc.numpy()
print("dense user0", ids[0][:8], sc[0][:8]); print("dense user7", ids[7][:8], sc[7][:8])
bad = [u for u in range(U) if u != 7 and (set(ids[u]) & {40, 41, 100})]
print("  dense users with NaN item:", bad[:5], ids.min(), ids.max(), sum(len(set(r)) != k for r in ids))
