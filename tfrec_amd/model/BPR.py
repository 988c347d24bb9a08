"""BPR-MF (Rendle et al.), with the reference's class shape (model/BPR.py:16-85).

Keys read from the configuration: the reference's (factors_num, lr, reg, epochs, batch_size, model/BPR.py:22-26)
and, all optional with the reference's behaviour as the default: ``sampler`` (host | device), ``seed``,
``item_mode`` (auto | dense | sharded: how the item table is laid out over several GPUs), ``checkpoint_file``,
``resume``.  Started as one process per GPU (run_model.py with ``devices = N``, or torchrun) the same class trains
on row-sharded tables (tfrec_amd/sharded.py): rank 0 logs, every rank steps.
"""
import numpy as np

from ..data import DataIterator, DeviceEpochSampler
from ..device import Device
from ..utils import csr_to_user_dict, random_choice
from .base import AbstractRecommender, MatrixFactorization


def process_group():
    """(rank, world) of the running process group; (0, 1) for a single process."""
    try:
        import torch.distributed as dist
    except ImportError:                  # pragma: no cover
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _Quiet:
    """Logger of the ranks other than 0."""

    def info(self, *a, **k):
        pass

    debug = warning = error = critical = info


class BPR(AbstractRecommender):
    n_neg = 1           # candidates drawn per positive (DNSBPR: neg_num)

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
        self.evaluator = evaluator
        self.sess = sess if sess is not None else Device(0)
        self.seed = int(config["seed"]) if "seed" in config else 2018
        # "sampler = device" (not a reference key): epochs are drawn and shuffled on the GPU; the default
        # keeps the reference's host sampler and its numpy random stream
        self.sampler = str(config["sampler"]).lower() if "sampler" in config else "host"
        if self.sampler not in ("host", "device"):
            raise ValueError("sampler must be 'host' or 'device'")
        self.rank, self.world = process_group()
        self.sharded = self.world > 1 or bool(config.get("force_sharded", False))
        self._ut = None
        if self.sharded:
            self._init_sharded(train_matrix)
            return
        self.user_pos_train = csr_to_user_dict(train_matrix)
        self.all_items = np.arange(self.items_num)
        self.mf = MatrixFactorization(self.sess, self.users_num, self.items_num, self.factors_num, name=self.__class__.__name__)
        self._device_sampler = self._make_device_sampler(train_matrix) if self.sampler == "device" else None

    def _make_device_sampler(self, train_matrix):
        return DeviceEpochSampler(self.sess, train_matrix, seed=self.seed, n_neg=self.n_neg)

    # -- several GPUs: row-sharded tables, one process per GPU ------------------------------------
    def _init_sharded(self, train_matrix):
        """Tables sharded by rows over the ranks of the process group, epochs drawn on every rank's GPU for the users
        it owns and cut into the batches ONE process would form (the draws and the shuffle do not depend on the
        number of ranks), evaluation sharded the way the item table is.  The host sampler of the reference is one
        numpy stream on one core and is not offered here."""
        from ..sharded import ShardedBprTrainer, ShardedEpochSampler
        if self.rank != 0:
            self.logger = _Quiet()
        if "sampler" in self.config and self.sampler == "host":
            raise ValueError("several GPUs draw their epochs on the device: set sampler = device (or drop the key)")
        self.trainer = ShardedBprTrainer(self.sess, self.users_num, self.items_num, self.factors_num, self.lr, self.reg, seed=self.seed,
                                         device=self._torch_device(), item_mode=str(self.config.get("item_mode", "auto")))
        self._epochs = ShardedEpochSampler(self.sess, self.trainer, train_matrix, seed=self.seed, n_neg=self.n_neg, shuffle="global")
        self._sharded_eval = None
        self._device_sampler = None

    def _torch_device(self):
        import torch
        # (a tfrec_amd.device.Device; tests inject a CPU stand-in for the local compute, which has no GPU)
        return torch.device("cuda", self.sess.device) if hasattr(self.sess, "lib") else torch.device("cpu")

    def evaluate_sharded(self, evaluator):
        """float32 [5 * top_k] on every rank (collective): the evaluator's users, lists and settings over the ranks."""
        import torch
        from ..sharded import ShardedEvaluator
        if self._sharded_eval is None:
            tr, te, users = evaluator.csr_arrays()
            dev = self.trainer.device
            up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)        # noqa: E731
            self._sharded_eval = ShardedEvaluator(self.sess, self.trainer, (up(tr[0], torch.int64), up(tr[1], torch.int32)),
                                                  (up(te[0], torch.int64), up(te[1], torch.int32)), up(users, torch.int32),
                                                  top_k=evaluator.top_k, mode=evaluator.score_mode)
        return self._sharded_eval.evaluate().cpu().numpy().reshape(-1)

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

    # batches from this size on run on a versioned user table (no staging of pre-step user rows) with the index of
    # batch k+1 sorted on the library's auxiliary stream beside the row passes of batch k; below it a step is bound
    # by its kernel launches and the plain table saves one
    VERSIONED_FROM = 1 << 16

    def _user_table(self):
        """The model's versioned user table, made on first use and kept: flattened, not destroyed, after an epoch."""
        if self._ut is None:
            self._ut = self.sess.user_table(self.mf.device_parameters()[0])
        return self._ut

    def _run_steps(self, du, dp, dn, bounds):
        """The minibatch steps of one epoch (model/BPR.py:68-72) over device arrays in batch order."""
        dev = self.sess
        P, Q, b = self.mf.device_parameters()
        n = len(bounds) - 1
        if self.batch_size < self.VERSIONED_FROM:
            for k in range(n):
                lo, hi = bounds[k], bounds[k + 1]
                dev.bpr_sgd_step(P, Q, b, du.view(lo, hi), dp.view(lo, hi), dn.view(lo, hi), self.lr, self.reg)
            dev.synchronize()
            return
        ut = self._user_table()
        U, I = self.users_num, self.items_num

        def prepare(k):
            lo, hi = bounds[k], bounds[k + 1]
            dev.prepare_batch(k & 1, U, I, du.view(lo, hi), dp.view(lo, hi), dn.view(lo, hi))
        try:
            if n > 0:
                prepare(0)
            for k in range(n):
                if k + 1 < n:
                    prepare(k + 1)
                dev.bpr_sgd_step_prepared_v(k & 1, ut, Q, b, self.lr, self.reg)
        finally:
            ut.flatten()            # P is complete again for the evaluation (or whatever else reads it next)
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
            if self.sharded:
                self._epochs.train_epoch(epoch, self.batch_size)
            elif self._device_sampler is not None:
                self._train_epoch_on_device(epoch)
            else:
                self._train_epoch(self.get_training_data())
            result = self.evaluate_model()
            self.logger.info("epoch %d:\t%s" % (epoch, result))
            self._save_checkpoint(self.config, epoch)

    def _checkpoint_arrays(self):
        if self.sharded:
            raise NotImplementedError("checkpoint_file / resume are single-GPU features; a sharded run saves with parameters()")
        return dict(zip(("P", "Q", "b"), self.mf.device_parameters()))

    def _uses_host_rng(self):
        return self._device_sampler is None

    def device_factors(self):
        return (self.sess,) + tuple(self.mf.device_parameters())

    def parameters(self):
        """Host copies of [P, Q, b]; with sharded tables assembled from the ranks (collective)."""
        if self.sharded:
            return [x.cpu().numpy() for x in self.trainer.gather_full()]
        return self.mf.parameters()

    def _predict(self):
        user_embedding, item_embedding, item_bias = self.parameters()
        return np.matmul(user_embedding, item_embedding.T) + item_bias

    def evaluate_model(self):
        result = self.evaluator.evaluate(self)
        return '\t'.join([str(x) for x in result])
