"""FoldOutEvaluator with the reference's interface (evaluator/evaluator.py:16-32)."""
import numpy as np
from scipy.sparse import csr_matrix

from ..utils.tools import typeassert, csr_to_user_dict
from .backend import eval_rating_matrix, eval_factors
from .backend.hip import _dict_to_csr


class AbstractEvaluator(object):
    def __init__(self):
        pass

    def evaluate(self, ranking_score):
        raise NotImplementedError


class FoldOutEvaluator(AbstractEvaluator):
    """Evaluator for the generic ranking task: every user with test items is ranked against
    all items except his train items; Precision/Recall/MAP/NDCG/MRR @1..top_k are averaged.

    Beside the reference's arguments (evaluator/evaluator.py:20): ``score_dtype`` "f32" (scores bit for bit the
    float32 fused-multiply-add chain; the default) or "bf16" (factors rounded to bfloat16, float32 accumulation: the
    fast matrix-core path); ``metric_sum`` "reference" (per-user float32 curves summed sequentially in user order on
    the host, bit for bit evaluate.h:40-92; the default) or "float64" (sums formed on the device in float64, no
    per-user table: means agree with the reference's to about 1e-6 relative)."""

    @typeassert(train_matrix=csr_matrix, test_matrix=csr_matrix, top_k=int)
    def __init__(self, train_matrix, test_matrix, top_k=50, score_dtype="f32", metric_sum="reference"):
        super(FoldOutEvaluator, self).__init__()
        if score_dtype not in ("f32", "bf16"):
            raise ValueError("score_dtype must be 'f32' or 'bf16'")
        if metric_sum not in ("reference", "float64"):
            raise ValueError("metric_sum must be 'reference' or 'float64'")
        self.top_k = top_k
        self.score_dtype, self.metric_sum = score_dtype, metric_sum
        self.user_pos_train = csr_to_user_dict(train_matrix)
        self.user_pos_test = csr_to_user_dict(test_matrix)
        self.num_users = train_matrix.shape[0]
        self._dev_cache = None
        self._csr_cache = None

    @classmethod
    def from_config(cls, config, train_matrix, test_matrix):
        """The evaluator run_model.py builds: the reference's (top 50, float32 scores, float32 means) unless the
        configuration carries ``top_k``, ``score_dtype`` or ``metric_sum``."""
        return cls(train_matrix, test_matrix, top_k=int(config.get("top_k", 50)), score_dtype=str(config.get("score_dtype", "f32")),
                   metric_sum=str(config.get("metric_sum", "reference")))

    @property
    def score_mode(self):
        from ..device import SCORE_BF16, SCORE_F32
        return SCORE_BF16 if self.score_dtype == "bf16" else SCORE_F32

    def metrics_info(self):
        return ", ".join("%s@1:1:%d" % (m, self.top_k) for m in ("Precision", "Recall", "MAP", "NDCG", "MRR"))

    def csr_arrays(self):
        """((train indptr, indices), (test indptr, indices), sorted test users) as numpy arrays over all users."""
        if self._csr_cache is None:
            tr = _dict_to_csr(self.user_pos_train, self.num_users)
            te = _dict_to_csr(self.user_pos_test, self.num_users)
            users = np.array(sorted(self.user_pos_test.keys()), dtype=np.int32)
            self._csr_cache = (tr, te, users)
        return self._csr_cache

    def _device_side(self, dev):
        """CSR structures and the sorted test-user list, uploaded once per device."""
        if self._dev_cache is None or self._dev_cache[0] is not dev:
            tr, te, users = self.csr_arrays()
            self._dev_cache = (dev, (dev.array(tr[0]), dev.array(tr[1])), (dev.array(te[0]), dev.array(te[1])), dev.array(users))
        return self._dev_cache[1:]

    def evaluate(self, model):
        """float32 [5*top_k].  A model that exposes its factors on the device (``device_factors()``
        returning (device, P, Q, b)) is evaluated in factor form; a model whose tables are sharded over several
        GPUs evaluates them in place (``evaluate_sharded``, collective, float64 sums); any other model goes through
        ``_predict()`` and the dense path, exactly like the reference (evaluator.py:29-32)."""
        mean_mode = 1 if self.metric_sum == "reference" else 0
        if getattr(model, "sharded", False):
            return model.evaluate_sharded(self)
        if hasattr(model, "device_factors"):
            dev, P, Q, b = model.device_factors()
            train, test, users = self._device_side(dev)
            result = eval_factors(P, Q, b, train, test, users, top_k=self.top_k, device=dev, mode=self.score_mode, mean_mode=mean_mode)
        else:
            ranking_score = model._predict()
            result = eval_rating_matrix(ranking_score, self.user_pos_train, self.user_pos_test, top_k=self.top_k, mean_mode=mean_mode)
        return result.flatten()
