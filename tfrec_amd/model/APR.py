"""Adversarial Personalized Ranking (He et al.), reference class shape: model/APR.py:17-195."""
import os

import numpy as np

from ..data import DataIterator
from ..device import Device
from ..utils import csr_to_user_dict, load_params, random_choice
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
        self.user_pos_train = csr_to_user_dict(train_matrix)
        self.all_items = np.arange(self.items_num)
        self.evaluator = evaluator
        self.sess = sess if sess is not None else Device(0)
        self._create_variables()

    def _create_variables(self):
        """Pretrained [P, Q, b] (model/APR.py:47-53), zero delta tables (:55-60) and Adagrad
        accumulators at TensorFlow's default initial value 0.1 (:136)."""
        path = self.pretrain_file
        if not os.path.exists(path) and os.path.exists(os.path.splitext(path)[0] + ".npz"):
            path = os.path.splitext(path)[0] + ".npz"
        params = load_params(path, trusted=self.trust_pretrain_pickle)
        dev = self.sess
        self.mf = MatrixFactorization(dev, self.users_num, self.items_num, self.factors_num, params=params, name="APR")
        shapes = [(self.users_num, self.factors_num), (self.items_num, self.factors_num), (self.items_num,)]
        self.deltas = [dev.zeros(s, np.float32) for s in shapes]
        self.accumulators = [dev.full(s, 0.1) for s in shapes]

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

    def train_model_epoch(self):
        """model/APR.py:156-164: per batch, the delta update and the Adagrad step (one library call)."""
        (users, pos, neg), bounds = self.get_train_data().columns_in_epoch_order()
        dev = self.sess
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
            self.train_model_epoch()
            result = self.evaluate_model()
            self.logger.info("%d:\t%s" % (epoch, result))
            self._save_checkpoint(self.config, epoch)

    def _checkpoint_arrays(self):
        """Parameters and Adagrad accumulators; the delta tables are rebuilt by every step."""
        out = dict(zip(("P", "Q", "b"), self.mf.device_parameters()))
        out.update(zip(("accP", "accQ", "accb"), self.accumulators))
        return out

    def device_factors(self):
        return (self.sess,) + tuple(self.mf.device_parameters())

    def evaluate_model(self):
        result = self.evaluator.evaluate(self)
        return '\t'.join([str(x) for x in result])

    def _predict(self):
        user_embedding, item_embedding, item_bias = self.mf.parameters()
        return np.matmul(user_embedding, item_embedding.T) + item_bias
