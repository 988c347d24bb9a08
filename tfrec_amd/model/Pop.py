"""Most-popular baseline with the reference's class shape (model/Pop.py:5-30): every user gets the
items ranked by how many training users interacted with them.  It has no factors, so it is the
client of the dense evaluator path (``_predict()`` -> eval_rating_matrix on the GPU)."""
import numpy as np

from .base import AbstractRecommender


class Pop(AbstractRecommender):
    def __init__(self, sess, config, dataset, evaluator):
        super(Pop, self).__init__(sess, config, dataset, evaluator)
        self.train_matrix = dataset.train_matrix
        self.users_num, self.items_num = self.train_matrix.shape
        self.evaluator = evaluator
        self.ranking_score = None

    def train_model(self):
        self.logger.info(self.evaluator.metrics_info())
        # stored interactions per item (model/Pop.py:14-15 counts the rows of the transposed matrix)
        csr = self.train_matrix.tocsr()
        self.ranking_score = np.bincount(csr.indices, minlength=self.items_num).astype(np.float32)
        self.logger.info(self.evaluator.metrics_info())
        result = self.evaluate_model()
        self.logger.info("result:\t%s" % result)

    def _predict(self):
        return np.ascontiguousarray(np.broadcast_to(self.ranking_score, (self.users_num, self.items_num)))

    def evaluate_model(self):
        result = self.evaluator.evaluate(self)
        return '\t'.join([str(x) for x in result])
