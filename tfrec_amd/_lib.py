"""ctypes binding of libtfrec_amd.so (the C ABI declared in include/tfrec_amd.h).

There is no CPU fallback: if the library is missing or does not load, importing the
product path fails with an ImportError that says how to build it.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# TFREC_AMD_LIB names another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("TFREC_AMD_LIB") or os.path.join(_HERE, "lib", "libtfrec_amd.so")

OK, EINVAL, EHIP, ENOMEM, EINDEX, EUNSUP = 0, -1, -2, -3, -4, -5
SCORE_F32, SCORE_BF16 = 0, 1

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_f = C.c_float
_u64 = C.c_uint64

# name -> (restype, argtypes); one entry per symbol declared in include/tfrec_amd.h
SIGNATURES = {
    "tfrec_last_error": (C.c_char_p, []),
    "tfrec_abi_version": (_int, []),
    "tfrec_ctx_create": (_int, [_int, _p, C.POINTER(_p)]),
    "tfrec_ctx_destroy": (_int, [_p]),
    "tfrec_ctx_set_stream": (_int, [_p, _p]),
    "tfrec_ctx_stream": (_p, [_p]),
    "tfrec_ctx_synchronize": (_int, [_p]),
    "tfrec_ctx_error_flags": (_int, [_p, C.POINTER(C.c_uint32), _int]),
    "tfrec_ctx_reserve": (_int, [_p, _i64, _int, _i64, _int]),
    "tfrec_ctx_workspace_bytes": (_i64, [_p]),
    "tfrec_ctx_set_profiling": (_int, [_p, _int]),
    "tfrec_ctx_phase_times": (_int, [_p, C.POINTER(C.c_float), _int, C.POINTER(_int)]),
    "tfrec_ctx_phase_name": (C.c_char_p, [_p, _int]),
    "tfrec_dev_malloc": (_int, [_p, C.c_size_t, C.POINTER(_p)]),
    "tfrec_dev_free": (_int, [_p, _p]),
    "tfrec_memcpy_h2d": (_int, [_p, _p, _p, C.c_size_t]),
    "tfrec_memcpy_d2h": (_int, [_p, _p, _p, C.c_size_t]),
    "tfrec_memset_zero": (_int, [_p, _p, C.c_size_t]),
    "tfrec_fill_uniform": (_int, [_p, _p, _i64, _f, _f, C.c_uint64]),
    "tfrec_fill_uniform_at": (_int, [_p, _p, _i64, _f, _f, C.c_uint64, _i64]),
    "tfrec_fill_const": (_int, [_p, _p, _i64, _f]),
    "tfrec_mf_predict": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _i64, _p]),
    "tfrec_sample_negatives": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _u64, _u64, _i64, _p, _p, _p]),
    "tfrec_shuffle_perm": (_int, [_p, _i64, _u64, _u64, _p]),
    "tfrec_epoch_triples": (_int, [_p, _p, _p, _i64, _i64, _i64, _u64, _u64, _u64, _i64, _p, _p, _p]),
    "tfrec_gather_rows_i32": (_int, [_p, _p, _i64, _p, _i64, _int, _p]),
    "tfrec_sort_pairs_u32": (_int, [_p, _p, _p, _p, _p, _i64, _int]),
    "tfrec_bpr_sgd_step": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_bpr_sgd_step_noalloc": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_batch_prepare": (_int, [_p, _int, _i64, _i64, _p, _p, _p, _i64]),
    "tfrec_batch_prepare_many": (_int, [_p, _int, _i64, _i64, _p, _p, _p, _i64, _int]),
    "tfrec_bpr_sgd_step_prepared": (_int, [_p, _int, _p, _p, _p, _i64, _i64, _int, _f, _f, _p]),
    "tfrec_utable_create": (_int, [_p, _p, _i64, _int, C.POINTER(_p)]),
    "tfrec_utable_flatten": (_int, [_p, _p]),
    "tfrec_utable_destroy": (_int, [_p, _p]),
    "tfrec_bpr_sgd_step_v": (_int, [_p, _p, _p, _p, _i64, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_bpr_sgd_step_prepared_v": (_int, [_p, _int, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_bpr_dense_grad_step_v": (_int, [_p, _int, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_bpr_dense_grad_step_chunked": (_int, [_p, _int, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _f, _f, _p, _int, _p]),
    "tfrec_bpr_grad_step": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_rows_sgd_apply": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _i64, _f]),
    "tfrec_bpr_grad_step_prepared": (_int, [_p, _int, _p, _p, _p, _i64, _i64, _int, _f, _f, _p]),
    "tfrec_bpr_grad_step_prepared_v": (_int, [_p, _int, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_items_dedup": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p, _p, _p, _p, _p]),
    "tfrec_rows_gather": (_int, [_p, _p, _p, _i64, _int, _p, _i64, _p, _p]),
    "tfrec_bpr_dense_grad_step": (_int, [_p, _int, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _i64, _f, _f, _p]),
    "tfrec_axpy": (_int, [_p, _p, _p, _f, _i64]),
    "tfrec_apr_adagrad_step": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _i64,
                                      _f, _f, _f, _f, _p]),
    "tfrec_apr_dense_delta_grads": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _i64, _f]),
    "tfrec_apr_delta_from_grads": (_int, [_p, _p, _p, _p, _i64, _int, _f]),
    "tfrec_apr_dense_adagrad_grads": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _i64, _i64, _f, _f, _f, _p]),
    "tfrec_adagrad_apply": (_int, [_p, _p, _p, _p, _f, _i64]),
    "tfrec_dns_select": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _i64, _int, _p]),
    "tfrec_score_topk": (_int, [_p, _p, _p, _p, _i64, _int, _i64, _i64, _p, _i64, _p, _p, _int, _int, _p, _p]),
    "tfrec_score_topk_shard": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _i64, _p, _p, _int, _int, _p, _p]),
    "tfrec_topk_merge": (_int, [_p, _p, _p, _int, _i64, _int, _p, _p]),
    "tfrec_topk_fill_masked": (_int, [_p, _p, _p, _i64, _int, _p, _p]),
    "tfrec_topk_dense": (_int, [_p, _p, _i64, _i64, _int, _p, _p, _p, _p, _p]),
    "tfrec_rank_metrics": (_int, [_p, _p, _p, _i64, _int, _p, _p, _int, _p, _p]),
    "tfrec_rank_metrics_sums": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _p]),
    "tfrec_eval_fullrank": (_int, [_p, _p, _p, _p, _i64, _int, _p, _i64, _p, _p, _p, _p, _int, _int, _int, _p, _p]),
}

_lib = None


def _share_torch_hip_runtime():
    """A process can drive the GPU through ONE HIP runtime only.  PyTorch-ROCm wheels bundle their own
    libamdhip64 (same soname as the system one): when torch is imported first, this library binds to that
    copy and both work; when this library came first it used to bind to the system runtime, and torch's
    later initialisation then found no GPU.  So if a torch with a bundled runtime is installed (imported
    or not -- it is not imported here), that runtime is loaded before ours, which makes the order of the
    imports irrelevant.  TFREC_AMD_SYSTEM_HIP=1 skips this."""
    if os.environ.get("TFREC_AMD_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Loads the shared library once and attaches the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C tfrec_amd/csrc -j8` (hipcc, gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ImportError("could not load %s: %s (a ROCm runtime is required)" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class TfrecError(RuntimeError):
    pass


_EXC = {EINVAL: ValueError, ENOMEM: MemoryError, EINDEX: IndexError, EUNSUP: NotImplementedError, EHIP: TfrecError}


def check(rc):
    if rc != OK:
        msg = load().tfrec_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, TfrecError)("tfrec_amd: %s (code %d)" % (msg, rc))
