"""CPU test: the shared library loads and exports every symbol include/tfrec_amd.h declares
(no compute call is made, so no GPU is needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tfrec_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfrec_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("tfrec_bpr_sgd_step", "tfrec_apr_adagrad_step", "tfrec_dns_select", "tfrec_score_topk",
                 "tfrec_topk_merge", "tfrec_topk_dense", "tfrec_rank_metrics", "tfrec_eval_fullrank"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "tfrec_amd", "lib", "libtfrec_amd.so")
    if not os.path.exists(path):
        pytest.fail("libtfrec_amd.so is not built: run `make -C tfrec_amd/csrc -j8`")
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.tfrec_abi_version.restype = ctypes.c_int
    assert lib.tfrec_abi_version() == 1


def test_python_binding_covers_the_header():
    from tfrec_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()
