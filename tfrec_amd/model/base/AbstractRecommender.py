"""The model plug-in contract (model/base/AbstractRecommender.py:1-17).  ``sess`` is a
tfrec_amd.device.Device (or None: one is created on GPU 0) where the reference passes a
tf.Session; the factory injects ``logger`` on the class."""


class AbstractRecommender(object):
    logger = None

    def __init__(self, sess, config, dataset, evaluator):
        pass

    def train_model(self):
        raise NotImplementedError

    def evaluate_model(self):
        raise NotImplementedError

    def _predict(self):
        """All users' rating matrix (dense); only sensible for small user x item counts."""
        raise NotImplementedError
