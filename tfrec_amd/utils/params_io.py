"""Reading and writing the [P, Q, b] parameter triple.

The reference saves it with pickle (model/DNSBPR.py:106-109) and loads it with pickle.load
(model/APR.py:47-53).  Files written by this package are .npz (preferred) or the same pickle of
three float32 arrays; a pickle that came from elsewhere (e.g. the reference's shipped
dataset/ml100k_saved_model.pkl) is read WITHOUT unpickling: the opcode stream is only
disassembled and the raw array bytes are lifted out of it.
"""
import pickle
import pickletools

import numpy as np


def save_params(path, P, Q, b):
    P, Q, b = [np.ascontiguousarray(x, dtype=np.float32) for x in (P, Q, b)]
    if path.endswith(".npz"):
        np.savez(path, P=P, Q=Q, b=b)
    else:
        with open(path, "wb") as fout:          # same container as the reference: a list of three arrays
            pickle.dump([P, Q, b], fout)


def read_param_pickle_safely(path):
    """Three float32 arrays from a pickle of [ndarray, ndarray, ndarray] without executing it."""
    with open(path, "rb") as fin:
        raw = fin.read()
    run, shape, blobs, dtypes = [], None, [], []
    # An ndarray's state tuple is (version, shape, dtype, is_fortran, DATA).  The payload is the bytes/str argument
    # that follows the is_fortran flag, possibly after _codecs.encode (named by a GLOBAL the first time, fetched from
    # the memo afterwards), which python-3 protocol-2 pickles wrap around a latin-1 text.  Its shape is the last
    # tuple of plain integers before it: the state tuple's second member (what lies between the two -- the dtype
    # and the flag -- holds no such tuple).  Protocol 5 writes numpy._core.numeric._frombuffer(bytearray, dtype,
    # shape, order) instead: there the payload comes first and the next tuple of plain integers is its shape.
    expect_data = await_shape = False
    payload_ops = ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8", "BINSTRING", "SHORT_BINSTRING", "BINUNICODE", "SHORT_BINUNICODE",
                   "BINUNICODE8")
    int_ops = ("BININT", "BININT1", "BININT2", "LONG1", "INT")
    # Protocols 4 and 5 name a global by the two texts on top of the stack before STACK_GLOBAL; either may come from
    # the memo.  Enough of the machine is followed to know them: the texts pushed, the memo slots that hold texts.
    top2, pushed, memo, n_memo = [], None, {}, 0
    frombuffer = False      # the next bytes / bytearray operand is the buffer argument of numpy's _frombuffer
    for op, arg, _ in pickletools.genops(raw):
        name = op.name
        if name in ("BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8"):
            pushed = str(arg)
            top2 = (top2 + [pushed])[-2:]
        elif name in ("BINGET", "LONG_BINGET"):
            pushed = memo.get(int(arg))
            if isinstance(pushed, tuple):            # a global fetched again (the second and third array reuse _frombuffer)
                frombuffer = pushed[1].endswith(" _frombuffer")
                top2 = []
            else:
                top2 = (top2 + [pushed])[-2:]
        elif name == "MEMOIZE":
            memo[n_memo] = pushed
            n_memo += 1
        elif name in ("BINPUT", "LONG_BINPUT"):
            memo[int(arg)] = pushed
        elif name != "STACK_GLOBAL":
            top2, pushed = [], None
        if name in ("GLOBAL", "STACK_GLOBAL"):
            glob = str(arg) if name == "GLOBAL" else (" ".join(top2) if len(top2) == 2 and None not in top2 else "?")
            top2, pushed = [], ("global", glob)
            if not (glob.startswith("numpy") or glob == "_codecs encode"):
                raise ValueError("%s references %r: not a plain numpy parameter pickle" % (path, glob))
            frombuffer = glob.endswith(" _frombuffer")
        if expect_data and name in payload_ops:
            blobs.append((arg.encode("latin-1") if isinstance(arg, str) else bytes(arg), shape))
            expect_data = False
            continue
        # protocol 5: the array's buffer in band -- a bytearray when the array was writable, bytes when it was not
        if name == "BYTEARRAY8" or (frombuffer and name in ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8")):
            blobs.append((bytes(arg), None))
            await_shape, run, frombuffer = True, [], False
            continue
        if name in ("NEWFALSE", "NEWTRUE"):
            expect_data = True
            run = []
            continue
        if not (expect_data and (name in ("BINGET", "LONG_BINGET", "MEMOIZE", "BINPUT", "LONG_BINPUT")
                                 or (name == "GLOBAL" and str(arg) == "_codecs encode"))):
            expect_data = False
        if name in int_ops:
            run.append(int(arg))
            continue
        if name in ("TUPLE1", "TUPLE2", "TUPLE3"):
            n = int(name[-1])
            if len(run) >= n:
                shape = tuple(run[-n:])
                if await_shape:
                    blobs[-1], await_shape = (blobs[-1][0], shape), False
        elif name in ("BINUNICODE", "SHORT_BINUNICODE") and arg in ("f4", "f8"):
            dtypes.append(arg)
        elif name == "SHORT_BINUNICODE" and arg == "F" and blobs and not await_shape:
            raise ValueError("%s: column-major arrays are not supported" % path)
        run = []
    if len(blobs) != 3:
        raise ValueError("%s: expected three array payloads, found %d" % (path, len(blobs)))
    item = 4 if (not dtypes or dtypes[0] == "f4") else 8
    out = []
    for blob, shp in blobs:
        if not shp or 0 in shp or int(np.prod(shp)) * item != len(blob):
            raise ValueError("%s: a payload of %d bytes does not fill its shape %r" % (path, len(blob), shp))
        out.append(np.frombuffer(blob, "<f%d" % item).reshape(shp).astype(np.float32))
    return out


def load_params(path, trusted=False):
    """[P, Q, b].  .npz is read with numpy; a pickle is unpickled only if ``trusted`` (a file this
    package wrote), otherwise it is parsed without execution."""
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        return [z["P"], z["Q"], z["b"]]
    if trusted:
        with open(path, "rb") as fin:
            return [np.asarray(x, dtype=np.float32) for x in pickle.load(fin)]
    return read_param_pickle_safely(path)


def save_checkpoint(path, arrays, epoch, rng_state=None):
    """Training state after ``epoch`` finished: named float32 arrays (P, Q, b and, for Adagrad models,
    their accumulators), the epoch number and, for the host sampler, numpy's legacy generator state
    (the device sampler needs none: its draws are a function of (seed, epoch, position)).  One .npz
    of plain arrays, written next to the target and renamed so that an interrupted write leaves the
    previous checkpoint intact."""
    import os
    out = {"arr_" + k: np.ascontiguousarray(v, dtype=np.float32) for k, v in arrays.items()}
    out["epoch"] = np.int64(epoch)
    if rng_state is not None:
        name, keys, pos, has_gauss, cached = rng_state
        if name != "MT19937":
            raise ValueError("only the legacy MT19937 state is stored")
        out["rng_keys"] = np.asarray(keys, dtype=np.uint32)
        out["rng_rest"] = np.array([pos, has_gauss], dtype=np.int64)
        out["rng_cached"] = np.float64(cached)
    tmp = path + ".tmp.npz"
    np.savez(tmp, **out)
    os.replace(tmp, path)


def load_checkpoint(path):
    """(arrays dict, epoch, rng_state or None) as written by ``save_checkpoint``."""
    z = np.load(path, allow_pickle=False)
    arrays = {k[4:]: z[k] for k in z.files if k.startswith("arr_")}
    rng = None
    if "rng_keys" in z.files:
        rest = z["rng_rest"]
        rng = ("MT19937", z["rng_keys"], int(rest[0]), int(rest[1]), float(z["rng_cached"]))
    return arrays, int(z["epoch"]), rng
