"""Entry point with the reference's shape (run_model.py:1-23): seed, device context, factory,
dataset, evaluator, model, train.  The device context stands where the reference opens a
tf.Session.

Not in the reference: with ``devices = N`` (N > 1) in tfrec.ini or the model's ini file this script starts N
processes of itself, one per GPU of the node (or is started that way by torchrun); the ranks join one process
group (RCCL), every model shards its tables over them, and rank 0 writes the log.  ``dist_backend = gloo`` with
``single_device = True`` runs all ranks on GPU 0 (testing).
"""
import os
import random
import sys

import numpy as np

np.random.seed(2018)
random.seed(2018)


def main(root="."):
    from tfrec_amd.model import ModelFactory
    factory = ModelFactory(root)
    devices = factory.devices()
    if devices > 1 and "WORLD_SIZE" not in os.environ:
        # before anything in this process touches the GPU; the children are fresh processes, nothing is re-executed
        from tfrec_amd.launch import launch_ranks
        return launch_ranks(devices, os.path.abspath(__file__), sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and devices not in (1, world):
        raise SystemExit("the configuration asks for %d devices, the launcher started %d ranks" % (devices, world))
    from tfrec_amd.data import Dataset
    from tfrec_amd.evaluator import FoldOutEvaluator
    if world > 1:
        import torch
        from tfrec_amd.launch import init_process_group
        from tfrec_amd.sharded import make_device
        opts = factory.options()
        local = 0 if opts.get("single_device", False) else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        init_process_group(str(opts.get("dist_backend", "nccl")), device=torch.device("cuda", local))
        sess = make_device(local)
    else:
        from tfrec_amd.device import Device
        sess = Device(0)
    Model, config = factory.get_model()
    dataset = Dataset(config)
    evaluator = FoldOutEvaluator.from_config(config, dataset.train_matrix, dataset.test_matrix)
    model = Model(sess, config, dataset, evaluator)
    model.train_model()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
