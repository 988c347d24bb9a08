"""The reference picks its C++ backend and falls back to pure Python if the extension is missing
(evaluator/backend/__init__.py:1-6).  Here there is one backend, on the GPU, and no fallback:
a missing library is an ImportError."""
from .hip import eval_rating_matrix, eval_factors
