"""The model plug-in contract (model/base/AbstractRecommender.py:1-17).  ``sess`` is a
tfrec_amd.device.Device (or None: one is created on GPU 0) where the reference passes a
tf.Session; the factory injects ``logger`` on the class.

Not in the reference: models that keep their state in device arrays can checkpoint after every
epoch and resume (config keys ``checkpoint_file`` and ``resume``; both optional, off by default)."""
import os

import numpy as np


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

    # -- checkpoint / resume ------------------------------------------------------------------
    def _checkpoint_arrays(self):
        """Name -> device array of everything an epoch changes (overridden by the models)."""
        return {}

    def _uses_host_rng(self):
        return True

    def _save_checkpoint(self, config, epoch):
        path = config.get("checkpoint_file") if hasattr(config, "get") else None
        if not path:
            return
        from ...utils import save_checkpoint
        arrays = {k: v.numpy() for k, v in self._checkpoint_arrays().items()}
        save_checkpoint(path, arrays, epoch, rng_state=np.random.get_state() if self._uses_host_rng() else None)

    def _restore_checkpoint(self, config):
        """Loads the checkpoint if ``resume`` is set and the file exists; returns the first epoch to run."""
        path = config.get("checkpoint_file") if hasattr(config, "get") else None
        if not path or not config.get("resume", False) or not os.path.exists(path):
            return 0
        from ...utils import load_checkpoint
        arrays, epoch, rng = load_checkpoint(path)
        mine = self._checkpoint_arrays()
        if set(arrays) != set(mine):
            raise ValueError("checkpoint %s holds %s, the model needs %s" % (path, sorted(arrays), sorted(mine)))
        for k, dst in mine.items():
            if tuple(arrays[k].shape) != tuple(dst.shape):
                raise ValueError("checkpoint array %s has shape %s, the model's is %s" % (k, arrays[k].shape, tuple(dst.shape)))
            dst.copy_from(arrays[k])
        if rng is not None and self._uses_host_rng():
            np.random.set_state(rng)
        if self.logger is not None:
            self.logger.info("resumed from %s after epoch %d" % (path, epoch))
        return epoch + 1
