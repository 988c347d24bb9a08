"""Not collected by pytest (run by hand on a GPU box): many epochs of the BPR and DNS-BPR plug-ins on a synthetic training
matrix with the epochs drawn on the device, small and large batches, an evaluation every few epochs -- free device memory
and the time per epoch must not drift.  python tests/soak_models.py [epochs]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tfrec_amd.device import Device  # noqa: E402
from tfrec_amd.evaluator import FoldOutEvaluator  # noqa: E402
from tfrec_amd.model import BPR, DNSBPR  # noqa: E402


class _Log:
    def info(self, *_):
        pass


class _Data:
    def __init__(self, U, I, per_user, rng):
        rows = np.repeat(np.arange(U), per_user)
        cols = np.concatenate([rng.choice(I, per_user, replace=False) for _ in range(U)])
        test = rng.random(len(rows)) < 0.1
        mk = lambda m: sp.csr_matrix((np.ones(m.sum(), np.float32), (rows[m], cols[m])), shape=(U, I))       # noqa: E731
        self.train_matrix, self.test_matrix = mk(~test), mk(test)
        self.train_matrix.sort_indices(); self.test_matrix.sort_indices()


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(3)
    data = _Data(20000, 5000, 50, rng)
    ev = FoldOutEvaluator(data.train_matrix, data.test_matrix, top_k=50)
    dev = Device(0)
    bad = 0
    for cls, batch, extra in ((BPR, 8192, {}), (BPR, 1 << 17, {}), (DNSBPR, 1 << 17, {"neg_num": 8, "data_name": "soak"})):
        cls.logger = _Log()
        cfg = dict({"factors_num": 64, "lr": 0.01, "reg": 0.05, "epochs": epochs, "batch_size": batch, "sampler": "device", "seed": 7}, **extra)
        m = cls(dev, cfg, data, ev)
        times, free = [], []
        for e in range(epochs):
            t0 = time.perf_counter()
            m._train_epoch_on_device(e)
            dev.synchronize()
            times.append(time.perf_counter() - t0)
            if e % 25 == 24:
                res = m.evaluate_model()
                free.append(torch.cuda.mem_get_info(0)[0])
        p10 = float(res.split("\t")[9])
        early, late = np.median(times[5:25]), np.median(times[-20:])
        drift = free[0] - free[-1]
        ok = late < 1.3 * early and drift <= (8 << 20) and np.isfinite(p10)
        bad += 0 if ok else 1
        print("%s batch %d: %d epochs, ms per epoch early %.2f late %.2f, free memory drift %.1f MiB, Precision@10 %.4f: %s" % (
            cls.__name__, batch, epochs, early * 1e3, late * 1e3, drift / 2 ** 20, p10, "ok" if ok else "DRIFT"), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
