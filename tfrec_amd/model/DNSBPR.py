"""BPR with dynamic negative sampling (model/DNSBPR.py:13-110)."""
import numpy as np

from ..data import DataIterator, DeviceEpochSampler
from ..utils import random_choice, save_params
from .BPR import BPR


class DNSBPR(BPR):
    def __init__(self, sess, config, dataset, evaluator):
        self.neg_num = config["neg_num"]
        self.n_neg = int(self.neg_num)        # pools of neg_num candidates per positive instead of one negative
        super(DNSBPR, self).__init__(sess, config, dataset, evaluator)
        self.config = config
        if self._device_sampler is not None:
            n = self._device_sampler.nnz
            self._picked = self.sess.empty(n, np.int32)
            self._picked_shuffled = self.sess.empty(n, np.int32)

    def _device_epoch(self, epoch):
        """The epoch of ``get_training_data`` without the host: candidate pools drawn on the GPU, the
        best-scored candidate of every pool under the current parameters, then the shuffle."""
        s, dev = self._device_sampler, self.sess
        users, pos, pools = s.epoch(epoch, shuffle=False)
        P, Q, b = self.mf.device_parameters()
        dev.dns_select(P, Q, b, users, pools, out=self._picked)
        perm = dev.shuffle_perm(s.nnz, s.seed, 2 * epoch + 1, out=s._perm)
        dev.gather_rows_i32(users, perm, out=s.users)
        dev.gather_rows_i32(pos, perm, out=s.pos)
        dev.gather_rows_i32(self._picked, perm, out=self._picked_shuffled)
        return s.users, s.pos, self._picked_shuffled

    def get_training_data(self):
        """For every positive draw ``neg_num`` uniform non-positives and keep the best-scored one
        under the CURRENT parameters (model/DNSBPR.py:65-80).  The candidate pools are drawn on
        the host, user after user in dict order (the order a single-worker executor gives the
        reference); the scoring and arg-max of all pools is one device call instead of one
        session call per user."""
        users, pos_items, pools = [], [], []
        for u, pos in self.user_pos_train.items():
            pool = random_choice(self.all_items, size=self.neg_num * len(pos), exclusion=pos)
            users.append(np.full(len(pos), u, dtype=np.int32))
            pos_items.append(np.asarray(pos, dtype=np.int32))
            pools.append(pool.reshape(len(pos), self.neg_num).astype(np.int32))
        users, pos_items, pools = np.concatenate(users), np.concatenate(pos_items), np.concatenate(pools)
        dev = self.sess
        P, Q, b = self.mf.device_parameters()
        neg = dev.dns_select(P, Q, b, dev.array(users), dev.array(pools))
        dev.synchronize()
        return DataIterator(users, pos_items, neg.numpy(), batch_size=self.batch_size, shuffle=True)

    def train_model(self):
        super(DNSBPR, self).train_model()
        self.save_model()

    def save_model(self):
        """The reference pickles [P, Q, b] to '<data_name>_dnsbpr.pkl' (model/DNSBPR.py:106-109);
        the same file is written here, plus an .npz twin."""
        P, Q, b = self.parameters()            # (sharded tables: assembled from the ranks, every rank takes part)
        if self.rank != 0:
            return
        save_params("%s_dnsbpr.pkl" % self.config["data_name"], P, Q, b)
        save_params("%s_dnsbpr.npz" % self.config["data_name"], P, Q, b)
