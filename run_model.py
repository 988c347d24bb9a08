"""Entry point with the reference's shape (run_model.py:1-23): seed, device context, factory,
dataset, evaluator, model, train.  The device context stands where the reference opens a
tf.Session."""
import random

import numpy as np

from tfrec_amd.data import Dataset
from tfrec_amd.device import Device
from tfrec_amd.evaluator import FoldOutEvaluator
from tfrec_amd.model import ModelFactory

np.random.seed(2018)
random.seed(2018)


if __name__ == "__main__":
    sess = Device(0)
    Model, config = ModelFactory().get_model()
    dataset = Dataset(config)
    evaluator = FoldOutEvaluator(dataset.train_matrix, dataset.test_matrix)
    model = Model(sess, config, dataset, evaluator)
    model.train_model()
