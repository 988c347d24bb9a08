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
    all items except his train items; Precision/Recall/MAP/NDCG/MRR @1..top_k are averaged."""

    @typeassert(train_matrix=csr_matrix, test_matrix=csr_matrix, top_k=int)
    def __init__(self, train_matrix, test_matrix, top_k=50):
        super(FoldOutEvaluator, self).__init__()
        self.top_k = top_k
        self.user_pos_train = csr_to_user_dict(train_matrix)
        self.user_pos_test = csr_to_user_dict(test_matrix)
        self.num_users = train_matrix.shape[0]
        self._dev_cache = None

    def metrics_info(self):
        return "Precision@1:1:50, Recall@1:1:50, MAP@1:1:50, NDCG@1:1:50, MRR@1:1:50"

    def _device_side(self, dev):
        """CSR structures and the sorted test-user list, uploaded once per device."""
        if self._dev_cache is None or self._dev_cache[0] is not dev:
            tr = _dict_to_csr(self.user_pos_train, self.num_users)
            te = _dict_to_csr(self.user_pos_test, self.num_users)
            users = np.array(sorted(self.user_pos_test.keys()), dtype=np.int32)
            self._dev_cache = (dev, (dev.array(tr[0]), dev.array(tr[1])), (dev.array(te[0]), dev.array(te[1])), dev.array(users))
        return self._dev_cache[1:]

    def evaluate(self, model):
        """float32 [5*top_k].  A model that exposes its factors on the device (``device_factors()``
        returning (device, P, Q, b)) is evaluated in factor form; any other model goes through
        ``_predict()`` and the dense path, exactly like the reference (evaluator.py:29-32)."""
        if hasattr(model, "device_factors"):
            dev, P, Q, b = model.device_factors()
            train, test, users = self._device_side(dev)
            result = eval_factors(P, Q, b, train, test, users, top_k=self.top_k, device=dev)
        else:
            ranking_score = model._predict()
            result = eval_rating_matrix(ranking_score, self.user_pos_train, self.user_pos_test, top_k=self.top_k)
        return result.flatten()
