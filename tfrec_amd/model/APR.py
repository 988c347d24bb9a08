"""Adversarial Personalized Ranking (He et al.), reference class shape: model/APR.py:17-195.

Optional keys beside the reference's (model/APR.py:26-30), all defaulting to the reference's behaviour: ``sampler``
(host | device), ``seed``, ``checkpoint_file``, ``resume``, ``trust_pretrain_pickle``.  One process per GPU
(run_model.py with ``devices = N``) trains on user tables sharded by rows (tfrec_amd/sharded.py, ShardedAprTrainer).
"""
import os

import numpy as np

from ..data import DataIterator, DeviceEpochSampler
from ..device import Device
from ..utils import csr_to_user_dict, load_params, random_choice
from .BPR import BPR, _Quiet, process_group
from .base import AbstractRecommender, MatrixFactorization


class APR(AbstractRecommender):
    def __init__(self, sess, config, dataset, evaluator):
        super(APR, self).__init__(sess, config, dataset, evaluator)
        self.config = config
        train_matrix = dataset.train_matrix
        self.users_num, self.items_num = train_matrix.shape
        self.factors_num = config["factors_num"]
        self.lr = config["lr"]
        self.reg = config["reg"]
        self.reg_adv = config["reg_adv"]
        self.epochs = config["epochs"]
        self.batch_size = config["batch_size"]
        self.eps = config["eps"]
        self.pretrain_file = config["pretrain_file"]
        self.trust_pretrain_pickle = bool(config.get("trust_pretrain_pickle", False))
        self.evaluator = evaluator
        self.sess = sess if sess is not None else Device(0)
        self.seed = int(config["seed"]) if "seed" in config else 2018
        # "sampler = device" (not a reference key): the epochs are drawn and shuffled on the GPU
        self.sampler = str(config["sampler"]).lower() if "sampler" in config else "host"
        if self.sampler not in ("host", "device"):
            raise ValueError("sampler must be 'host' or 'device'")
        self.rank, self.world = process_group()
        self.sharded = self.world > 1 or bool(config.get("force_sharded", False))
        self._device_sampler = None
        if self.sharded:
            self._init_sharded(train_matrix)
            return
        self.user_pos_train = csr_to_user_dict(train_matrix)
        self.all_items = np.arange(self.items_num)
        self._create_variables()
        if self.sampler == "device":
            self._device_sampler = DeviceEpochSampler(self.sess, train_matrix, seed=self.seed)

    def _load_pretrained(self):
        path = self.pretrain_file
        if not os.path.exists(path) and os.path.exists(os.path.splitext(path)[0] + ".npz"):
            path = os.path.splitext(path)[0] + ".npz"
        return load_params(path, trusted=self.trust_pretrain_pickle)

    def _create_variables(self):
        """Pretrained [P, Q, b] (model/APR.py:47-53), zero delta tables (:55-60) and Adagrad
        accumulators at TensorFlow's default initial value 0.1 (:136)."""
        params = self._load_pretrained()
        dev = self.sess
        self.mf = MatrixFactorization(dev, self.users_num, self.items_num, self.factors_num, params=params, name="APR")
        shapes = [(self.users_num, self.factors_num), (self.items_num, self.factors_num), (self.items_num,)]
        self.deltas = [dev.zeros(s, np.float32) for s in shapes]
        self.accumulators = [dev.full(s, 0.1) for s in shapes]

    # -- several GPUs: user tables sharded by rows, item tables replicated, two all-reduces per step --
    def _init_sharded(self, train_matrix):
        from ..sharded import ShardedAprTrainer, ShardedEpochSampler
        if self.rank != 0:
            self.logger = _Quiet()
        if "sampler" in self.config and self.sampler == "host":
            raise ValueError("several GPUs draw their epochs on the device: set sampler = device (or drop the key)")
        self.trainer = ShardedAprTrainer(self.sess, self.users_num, self.items_num, self.factors_num, self.lr, self.reg, self.reg_adv,
                                         self.eps, params=self._load_pretrained(), seed=self.seed, device=BPR._torch_device(self))
        self._epochs = ShardedEpochSampler(self.sess, self.trainer, train_matrix, seed=self.seed, n_neg=1, shuffle="global")
        self._sharded_eval = None

    evaluate_sharded = BPR.evaluate_sharded

    def get_train_data(self):
        """model/APR.py:166-184: the same uniform sampler as BPR, users in dict order."""
        users, pos_items, neg_items = [], [], []
        for u, pos in self.user_pos_train.items():
            neg = random_choice(self.all_items, size=len(pos), exclusion=pos)
            users.append(np.full(len(pos), u, dtype=np.int32))
            pos_items.append(np.asarray(pos, dtype=np.int32))
            neg_items.append(neg.astype(np.int32))
        return DataIterator(np.concatenate(users), np.concatenate(pos_items), np.concatenate(neg_items),
                            batch_size=self.batch_size, shuffle=True)

    def train_model_epoch(self, epoch=0):
        """model/APR.py:156-164: per batch, the delta update and the Adagrad step (one library call)."""
        dev = self.sess
        if self.sharded:
            self._epochs.train_epoch(epoch, self.batch_size)
            return
        if self._device_sampler is not None:
            du, dp, dn = self._device_sampler.epoch(epoch)
            n = self._device_sampler.nnz
            bounds = list(range(0, n, self.batch_size)) + [n]
        else:
            (users, pos, neg), bounds = self.get_train_data().columns_in_epoch_order()
            du, dp, dn = dev.array(users, np.int32), dev.array(pos, np.int32), dev.array(neg, np.int32)
        P, Q, b = self.mf.device_parameters()
        for k in range(len(bounds) - 1):
            lo, hi = bounds[k], bounds[k + 1]
            dev.apr_adagrad_step(P, Q, b, *self.accumulators, *self.deltas, du.view(lo, hi), dp.view(lo, hi), dn.view(lo, hi),
                                 self.lr, self.reg, self.reg_adv, self.eps)
        dev.synchronize()

    def train_model(self):
        self.logger.info(self.evaluator.metrics_info())
        result = self.evaluate_model()
        self.logger.info("pretrain:\t%s" % result)
        for epoch in range(self._restore_checkpoint(self.config), self.epochs):
            self.train_model_epoch(epoch)
            result = self.evaluate_model()
            self.logger.info("%d:\t%s" % (epoch, result))
            self._save_checkpoint(self.config, epoch)

    def _checkpoint_arrays(self):
        """Parameters and Adagrad accumulators; the delta tables are rebuilt by every step."""
        if self.sharded:
            raise NotImplementedError("checkpoint_file / resume are single-GPU features")
        out = dict(zip(("P", "Q", "b"), self.mf.device_parameters()))
        out.update(zip(("accP", "accQ", "accb"), self.accumulators))
        return out

    def _uses_host_rng(self):
        return self._device_sampler is None

    def device_factors(self):
        return (self.sess,) + tuple(self.mf.device_parameters())

    def parameters(self):
        """Host copies of [P, Q, b]; with sharded user tables assembled from the ranks (collective)."""
        if self.sharded:
            return [x.cpu().numpy() for x in self.trainer.gather_full()]
        return self.mf.parameters()

    def evaluate_model(self):
        result = self.evaluator.evaluate(self)
        return '\t'.join([str(x) for x in result])

    def _predict(self):
        user_embedding, item_embedding, item_bias = self.parameters()
        return np.matmul(user_embedding, item_embedding.T) + item_bias
