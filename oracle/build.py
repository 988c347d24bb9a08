"""TEST INFRASTRUCTURE ONLY: build recipe for the CPU oracle.

Two artefacts:

* ``oracle/libmf_oracle.so`` -- this repository's own C restatement
  (``oracle/mf_oracle.c``); always built.
* ``oracle/_ref/libref_eval.so`` and ``oracle/_ref/apt_evaluate*.so`` -- the
  reference's evaluator compiled from its sources where they lie (header-only
  C++ under ``evaluator/backend/cpp`` plus its Cython shim); built only where the
  reference tree is mounted.  Nothing is copied out of it: the compiler reads the
  headers through ``-I`` and Cython's generated C++ lands in ``oracle/_ref/``
  (git-ignored, but shipped to the GPU box like any other built library).

The reference's own build (``setup.py build_ext --inplace``) is not run: it
writes next to the sources, which are read-only here.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("TFREC_REFERENCE", "/root/reference")
REF_CPP = os.path.join(REF_ROOT, "evaluator", "backend", "cpp")
REF_OUT = os.path.join(HERE, "_ref")


def _newer(target, *sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def build_oracle(force=False):
    src = os.path.join(HERE, "mf_oracle.c")
    out = os.path.join(HERE, "libmf_oracle.so")
    if not force and _newer(out, src):
        return out
    # -ffp-contract=off: every rounding in the restatement is the one written in the source
    # -fopenmp: only oracle_bpr_step_mt (the multi-core CPU baseline of the benchmark) has parallel regions
    cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", src, "-o", out, "-lm"]
    subprocess.check_call(cmd)
    return out


def reference_available():
    return os.path.isfile(os.path.join(REF_CPP, "evaluate.h"))


def build_reference(force=False):
    """Compile the reference evaluator in place; returns the output dir or None."""
    if not reference_available():
        return REF_OUT if os.path.isdir(REF_OUT) else None
    os.makedirs(REF_OUT, exist_ok=True)
    harness = os.path.join(HERE, "ref_harness.cpp")
    lib = os.path.join(REF_OUT, "libref_eval.so")
    hdrs = [os.path.join(REF_CPP, h) for h in ("tools.h", "metrics.h", "evaluate.h", "thread_pool.h")]
    if force or not _newer(lib, harness, *hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++11", "-fPIC", "-shared", "-pthread", "-I", REF_CPP,
                               harness, "-o", lib])
    # the reference's Cython shim, cythonized from its own .pyx
    pyx = os.path.join(REF_CPP, "apt_evaluate.pyx")
    gen = os.path.join(REF_OUT, "apt_evaluate.cpp")
    ext = os.path.join(REF_OUT, "apt_evaluate" + sysconfig.get_config_var("EXT_SUFFIX"))
    if force or not _newer(ext, pyx, *hdrs):
        import numpy
        subprocess.check_call([sys.executable, "-m", "cython", "-3", "--cplus", pyx, "-o", gen])
        subprocess.check_call(["g++", "-O2", "-std=c++11", "-fPIC", "-shared", "-pthread", "-w",
                               "-I", REF_CPP, "-I", sysconfig.get_paths()["include"], "-I", numpy.get_include(),
                               gen, "-o", ext])
    return REF_OUT


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    print(build_reference(force="--force" in sys.argv))
