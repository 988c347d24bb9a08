"""MI355X-native matrix-factorization ranking path with the TFRec plug-in surface.

Layout mirrors the reference's packages for this path only:
  tfrec_amd.model      BPR / DNSBPR / APR recommenders, MatrixFactorization, ModelFactory
  tfrec_amd.evaluator  FoldOutEvaluator, eval_rating_matrix
  tfrec_amd.data       Dataset, DataIterator
  tfrec_amd.utils      random_choice, csr_to_user_dict, typeassert, timer, Logger
  tfrec_amd.device     the device context (ctypes over include/tfrec_amd.h)
  tfrec_amd.sharded    multi-GPU training / evaluation over torch.distributed (RCCL)
"""
__version__ = "0.1.0"
