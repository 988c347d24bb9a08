"""BPR-MF (Rendle et al.), with the reference's class shape (model/BPR.py:16-85)."""
import numpy as np

from ..data import DataIterator, DeviceEpochSampler
from ..device import Device
from ..utils import csr_to_user_dict, random_choice
from .base import AbstractRecommender, MatrixFactorization


class BPR(AbstractRecommender):
    def __init__(self, sess, config, dataset, evaluator):
        super(BPR, self).__init__(sess, config, dataset, evaluator)
        self.config = config
        train_matrix = dataset.train_matrix
        self.users_num, self.items_num = train_matrix.shape
        self.factors_num = config["factors_num"]
        self.lr = config["lr"]
        self.reg = config["reg"]
        self.epochs = config["epochs"]
        self.batch_size = config["batch_size"]
        self.user_pos_train = csr_to_user_dict(train_matrix)
        self.all_items = np.arange(self.items_num)
        self.evaluator = evaluator
        self.sess = sess if sess is not None else Device(0)
        self.mf = MatrixFactorization(self.sess, self.users_num, self.items_num, self.factors_num, name=self.__class__.__name__)
        # "sampler = device" (not a reference key): epochs are drawn and shuffled on the GPU; the default
        # keeps the reference's host sampler and its numpy random stream
        self.sampler = str(config["sampler"]).lower() if "sampler" in config else "host"
        if self.sampler not in ("host", "device"):
            raise ValueError("sampler must be 'host' or 'device'")
        self._device_sampler = (DeviceEpochSampler(self.sess, train_matrix, seed=int(config["seed"]) if "seed" in config else 2018)
                                if self.sampler == "device" else None)

    # -- sampling: uniform non-positive per positive, users in dict order (model/BPR.py:50-62) --
    def get_training_data(self):
        users, pos_items, neg_items = [], [], []
        for u, pos in self.user_pos_train.items():
            neg = random_choice(self.all_items, size=len(pos), exclusion=pos)
            users.append(np.full(len(pos), u, dtype=np.int32))
            pos_items.append(np.asarray(pos, dtype=np.int32))
            neg_items.append(neg.astype(np.int32))
        return DataIterator(np.concatenate(users), np.concatenate(pos_items), np.concatenate(neg_items),
                            batch_size=self.batch_size, shuffle=True)

    # batches from this size on run on a versioned user table (no staging of pre-step user rows); below it a step
    # is bound by its kernel launches and the plain table saves one
    VERSIONED_FROM = 1 << 16

    def _run_steps(self, du, dp, dn, bounds):
        """The minibatch steps of one epoch (model/BPR.py:68-72) over device arrays in batch order."""
        dev = self.sess
        P, Q, b = self.mf.device_parameters()
        ut = dev.user_table(P) if self.batch_size >= self.VERSIONED_FROM else None
        for k in range(len(bounds) - 1):
            lo, hi = bounds[k], bounds[k + 1]
            if ut is not None:
                dev.bpr_sgd_step_v(ut, Q, b, du.view(lo, hi), dp.view(lo, hi), dn.view(lo, hi), self.lr, self.reg)
            else:
                dev.bpr_sgd_step(P, Q, b, du.view(lo, hi), dp.view(lo, hi), dn.view(lo, hi), self.lr, self.reg)
        if ut is not None:
            ut.close()              # P is complete again for the evaluation that follows
        dev.synchronize()

    def _train_epoch(self, train_data):
        """One epoch of minibatch steps (model/BPR.py:68-72): the epoch's triples are uploaded once
        in batch order and every batch is one step on a slice of them."""
        (users, pos, neg), bounds = train_data.columns_in_epoch_order()
        dev = self.sess
        self._run_steps(dev.array(users, np.int32), dev.array(pos, np.int32), dev.array(neg, np.int32), bounds)

    def _device_epoch(self, epoch):
        """Device arrays (users, pos, neg) of the epoch, shuffled."""
        return self._device_sampler.epoch(epoch)

    def _train_epoch_on_device(self, epoch):
        """The same epoch with the triples drawn, shuffled and consumed on the GPU."""
        users, pos, neg = self._device_epoch(epoch)
        n = self._device_sampler.nnz
        self._run_steps(users, pos, neg, list(range(0, n, self.batch_size)) + [n])

    def train_model(self):
        self.logger.info(self.evaluator.metrics_info())
        for epoch in range(self._restore_checkpoint(self.config), self.epochs):
            if self._device_sampler is not None:
                self._train_epoch_on_device(epoch)
            else:
                self._train_epoch(self.get_training_data())
            result = self.evaluate_model()
            self.logger.info("epoch %d:\t%s" % (epoch, result))
            self._save_checkpoint(self.config, epoch)

    def _checkpoint_arrays(self):
        return dict(zip(("P", "Q", "b"), self.mf.device_parameters()))

    def _uses_host_rng(self):
        return self._device_sampler is None

    def device_factors(self):
        return (self.sess,) + tuple(self.mf.device_parameters())

    def _predict(self):
        user_embedding, item_embedding, item_bias = self.mf.parameters()
        return np.matmul(user_embedding, item_embedding.T) + item_bias

    def evaluate_model(self):
        result = self.evaluator.evaluate(self)
        return '\t'.join([str(x) for x in result])
