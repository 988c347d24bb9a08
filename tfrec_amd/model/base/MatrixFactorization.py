"""Matrix-factorization parameters resident on the GPU (model/base/MatrixFactorization.py:4-78).

The reference builds TensorFlow variables and graph fragments; here the class owns three device
arrays and the operations are calls into the HIP library:
  user_embeddings [U,d], item_embeddings [I,d] ~ U(-0.05, 0.05), item_biases [I] = 0
  (MatrixFactorization.py:17-21), or the three arrays passed as ``params`` (:12-15).
"""
import numpy as np


class MatrixFactorization(object):
    def __init__(self, dev, users_num, items_num, factors_num, params=None, name="MF", seed=2018):
        self.dev = dev
        self.users_num, self.items_num, self.factors_num = int(users_num), int(items_num), int(factors_num)
        self.name = name
        if params is not None:
            P, Q, b = [np.ascontiguousarray(x, dtype=np.float32) for x in params]
            if P.shape != (self.users_num, self.factors_num) or Q.shape != (self.items_num, self.factors_num) or b.shape != (self.items_num,):
                raise ValueError("parameter shapes %s %s %s do not match the model" % (P.shape, Q.shape, b.shape))
            self.user_embeddings, self.item_embeddings, self.item_biases = dev.array(P), dev.array(Q), dev.array(b)
        else:
            self.user_embeddings = dev.uniform((self.users_num, self.factors_num), -0.05, 0.05, seed * 2 + 1)
            self.item_embeddings = dev.uniform((self.items_num, self.factors_num), -0.05, 0.05, seed * 2 + 2)
            self.item_biases = dev.zeros(self.items_num, np.float32)

    def device_parameters(self):
        return [self.user_embeddings, self.item_embeddings, self.item_biases]

    def parameters(self):
        """Host copies, in the order the reference returns them (MatrixFactorization.py:23-24)."""
        return [a.numpy() for a in self.device_parameters()]

    def predict(self, users, items):
        """y(u,i) = <P[u], Q[i]> + b[i] for parallel index arrays (MatrixFactorization.py:26-32)."""
        du, di = self.dev.array(users, np.int32), self.dev.array(items, np.int32)
        out = self.dev.predict(self.user_embeddings, self.item_embeddings, self.item_biases, du, di)
        self.dev.synchronize()
        return out.numpy()

    def all_logits(self, users=None):
        """Scores of the given users (or all) against all items (MatrixFactorization.py:34-46),
        on the host: meant for small problems and debugging, not for the training path."""
        P, Q, b = self.parameters()
        rows = P if users is None else P[np.asarray(users)]
        return np.matmul(rows, Q.T) + b
