"""Developer probe: wall-clock of the reference's own configuration (ML-100k, BPR, d=64, batches of 512) end to end:
per epoch the sampler + the minibatch steps, and the full-rank evaluation; host sampler and device sampler."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tfrec_amd.data import Dataset  # noqa: E402
from tfrec_amd.device import Device  # noqa: E402
from tfrec_amd.evaluator import FoldOutEvaluator  # noqa: E402
from tfrec_amd.model import BPR  # noqa: E402


class _Log:
    def info(self, *_):
        pass


def main():
    ds = Dataset({"data_name": "ml100k", "train_file": os.path.join(ROOT, "dataset", "movielens-100k-train.txt"),
                  "test_file": os.path.join(ROOT, "dataset", "movielens-100k-test.txt"), "format": "UIR", "separator": "\t"})
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=50)
    dev = Device(0)
    for sampler in ("host", "device"):
        np.random.seed(2018)
        BPR.logger = _Log()
        cfg = {"factors_num": 64, "lr": 0.01, "reg": 0.05, "epochs": 5, "batch_size": 512, "sampler": sampler, "seed": 2018}
        m = BPR(dev, cfg, ds, ev)
        t_train, t_eval = [], []
        for e in range(5):
            t0 = time.perf_counter()
            if sampler == "device":
                m._train_epoch_on_device(e)
            else:
                m._train_epoch(m.get_training_data())
            dev.synchronize()
            t1 = time.perf_counter()
            res = m.evaluate_model()
            t2 = time.perf_counter()
            t_train.append(t1 - t0); t_eval.append(t2 - t1)
        p10 = float(res.split("\t")[9])
        print(f"sampler={sampler}: epoch (sampling + {int(np.ceil(ds.train_matrix.nnz / 512))} steps) {min(t_train)*1e3:.1f} ms, "
              f"evaluation {min(t_eval)*1e3:.1f} ms, Precision@10 after 5 epochs {p10:.4f}", flush=True)


if __name__ == "__main__":
    main()
