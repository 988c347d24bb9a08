"""Generates the golden vectors under tests/golden/ (run in the build container only).

It needs the reference tree (read-only, default /root/reference) and the
reference evaluator built by ``oracle/build.py`` into ``oracle/_ref``.  The
reference itself never travels: only the inputs and the outputs it produced are
committed, as .npz files, next to this script.

What produces what
------------------
* evaluator vectors      -> the reference's own ``evaluator.FoldOutEvaluator`` /
  ``eval_rating_matrix`` (evaluator/evaluator.py:16-32,
  evaluator/backend/cpp/evaluate.py:10-44) running on the reference's own C++
  (apt_evaluate.pyx cythonized in place); rank lists from the same C++ through
  oracle/ref_harness.cpp.
* sampler vectors        -> the reference's ``utils.random_choice``
  (utils/tools.py:39-49), ``utils.csr_to_user_dict`` (utils/tools.py:52-62) and
  ``data.DataIterator`` (data/dataiterator.py:158-192), imported from the tree.
* splitter vectors       -> the reference's ``data/utils.py`` functions (data/utils.py:7-98), imported
  from the tree (``--only-splitter`` regenerates just these).
* training-step vectors  -> NOT the reference (TensorFlow is not installed):
  an independent float64 torch-autograd implementation of the graphs in
  model/BPR.py:36-48 and model/APR.py:64-136.  Parity with TensorFlow itself
  stays unpinned; these vectors pin the oracle and the HIP path to each other
  and to autograd.
"""
import hashlib
import importlib
import os
import pickletools
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TFREC_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build as oracle_build  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def import_reference():
    """Make the reference's pure-Python packages importable, with its Cython module
    (built from its own .pyx into oracle/_ref) standing where its package expects it."""
    refdir = oracle_build.build_reference()
    sys.path.insert(0, refdir)
    apt = importlib.import_module("apt_evaluate")
    sys.modules["evaluator.backend.cpp.apt_evaluate"] = apt
    sys.path.insert(0, REF)
    import utils  # noqa
    import data  # noqa
    import evaluator  # noqa
    from evaluator.backend.cpp.evaluate import eval_rating_matrix
    assert evaluator.evaluator.eval_rating_matrix is eval_rating_matrix, "cpp backend not selected"
    return utils, data, evaluator


def read_saved_model(path):
    """Pull the three float32 arrays out of the reference's pickle WITHOUT unpickling it:
    pickletools.genops only disassembles.  Layout seen in the opcode stream: a list of three
    C-ordered '<f4' ndarrays whose raw bytes are the three BINBYTES arguments."""
    with open(path, "rb") as f:
        raw = f.read()
    shapes, blobs, ints = [], [], []
    for op, arg, _ in pickletools.genops(raw):
        if op.name in ("BININT", "BININT1", "BININT2"):
            ints.append(arg)
        elif op.name in ("TUPLE1", "TUPLE2"):
            n = 1 if op.name == "TUPLE1" else 2
            shapes.append(tuple(ints[-n:]))
        elif op.name == "BINBYTES":
            blobs.append(arg)
        elif op.name == "GLOBAL":
            assert arg.split()[0].startswith("numpy"), arg
    out = []
    for blob in blobs:
        n = len(blob) // 4
        shp = [s for s in shapes if int(np.prod(s)) == n and s != (0,)][0]
        out.append(np.frombuffer(blob, "<f4").reshape(shp).copy())
    assert [a.shape for a in out] == [(943, 20), (1682, 20), (1682,)], [a.shape for a in out]
    return out


def pairs(csr):
    coo = csr.tocoo()
    order = np.lexsort((coo.col, coo.row))
    return coo.row[order].astype(np.uint16), coo.col[order].astype(np.uint16)


def margin_ok(R, ranks_k1, rel=2e-4):
    """True if the top-(k+1) of every row is separated by more than rel*|score|."""
    for r in range(R.shape[0]):
        s = R[r, ranks_k1[r]]
        s = s[np.isfinite(s)]
        if len(s) > 1 and np.min(-np.diff(s)) <= rel * np.max(np.abs(s)):
            return False
    return True


def gen_eval(utils, data, evaluator, lib):
    from evaluator.backend.cpp.evaluate import eval_rating_matrix
    out = {}
    # ---- (1) known answer: shipped parameters on the shipped ML-100k split -------------
    cfg = {"data_name": "ml100k", "train_file": os.path.join(REF, "dataset/movielens-100k-train.txt"),
           "test_file": os.path.join(REF, "dataset/movielens-100k-test.txt"), "format": "UIR", "separator": "\t"}
    ds = data.Dataset(cfg)
    P, Q, b = read_saved_model(os.path.join(REF, "dataset/ml100k_saved_model.pkl"))

    class Model:
        def _predict(self):
            return np.matmul(P, Q.T) + b          # model/BPR.py:78-79

    ev = evaluator.FoldOutEvaluator(ds.train_matrix, ds.test_matrix)
    res = ev.evaluate(Model()).reshape(5, -1)
    print("ML-100k known answer P@1/10/50:", res[0, [0, 9, 49]], "NDCG@50", res[3, 49])
    assert abs(res[0, 0] - 0.61220044) < 1e-7 and abs(res[3, 49] - 0.33114389) < 1e-7
    R = (np.matmul(P, Q.T) + b).astype(np.float32)
    for u, items in ev.user_pos_train.items():
        R[u][items] = -np.inf
    users = sorted(ev.user_pos_test.keys())
    ranks = lib.ref_topk_rows(R[users], 50)
    ranks51 = lib.ref_topk_rows(R[users], 51)
    tr_u, tr_i = pairs(ds.train_matrix)
    te_u, te_i = pairs(ds.test_matrix)
    np.savez_compressed(os.path.join(HERE, "ml100k_known_answer.npz"),
                        P=P, Q=Q, b=b, train_u=tr_u, train_i=tr_i, test_u=te_u, test_i=te_i,
                        metrics=res, ranks=ranks.astype(np.uint16), test_users=np.array(users, np.int32),
                        tie_free=np.array(margin_ok(R[users], ranks51, rel=1e-5)))
    out["ml100k"] = (ds, ev)

    # ---- (2) random tie-free factor models through the reference evaluator ------------
    rng = np.random.default_rng(2018)
    cases = {}

    def run_case(name, U, I, d, k_list, train_deg, test_deg, test_frac=1.0, scale=0.5):
        Pm = rng.uniform(-scale, scale, (U, d)).astype(np.float32)
        Qm = rng.uniform(-scale, scale, (I, d)).astype(np.float32)
        bm = rng.uniform(-0.1, 0.1, I).astype(np.float32)
        tr_rows, tr_cols, te_rows, te_cols = [], [], [], []
        for u in range(U):
            ntr = int(rng.integers(train_deg[0], train_deg[1] + 1))
            nte = int(rng.integers(test_deg[0], test_deg[1] + 1)) if rng.random() < test_frac else 0
            picks = rng.permutation(I)[:ntr + nte]
            tr_rows += [u] * ntr; tr_cols += picks[:ntr].tolist()
            te_rows += [u] * nte; te_cols += picks[ntr:].tolist()
        tr = sp.csr_matrix((np.ones(len(tr_rows)), (tr_rows, tr_cols)), shape=(U, I))
        te = sp.csr_matrix((np.ones(len(te_rows)), (te_rows, te_cols)), shape=(U, I))
        tr.sort_indices(); te.sort_indices()
        e = evaluator.FoldOutEvaluator(tr, te, top_k=max(k_list))
        tus = sorted(e.user_pos_test.keys())
        kk = min(max(k_list) + 1, I)
        # re-draw a user's factors until its top-(k+1) is separated by a clear margin, so that the
        # rank lists do not depend on the summation order of whoever computes the scores
        for u in tus:
            for attempt in range(1000):
                row = (np.matmul(Pm[u:u + 1], Qm.T) + bm).astype(np.float32)
                if u in e.user_pos_train:
                    row[0][e.user_pos_train[u]] = -np.inf
                if margin_ok(row, lib.ref_topk_rows(row, kk), rel=2e-5):
                    break
                Pm[u] = rng.uniform(-scale, scale, d).astype(np.float32)
            else:
                raise RuntimeError("no margin for user %d" % u)
        Rm = (np.matmul(Pm, Qm.T) + bm).astype(np.float32)
        Rmask = Rm.copy()
        for u, items in e.user_pos_train.items():
            Rmask[u][items] = -np.inf
        assert margin_ok(Rmask[tus], lib.ref_topk_rows(Rmask[tus], kk), rel=1e-5), name
        d_out = {"P": Pm, "Q": Qm, "b": bm, "train_indptr": tr.indptr.astype(np.int64),
                 "train_indices": tr.indices.astype(np.int32), "test_indptr": te.indptr.astype(np.int64),
                 "test_indices": te.indices.astype(np.int32), "test_users": np.array(tus, np.int32),
                 "k_list": np.array(k_list, np.int32)}
        for k in k_list:
            ek = evaluator.FoldOutEvaluator(tr, te, top_k=k)

            class M:
                def _predict(self):
                    return np.matmul(Pm, Qm.T) + bm
            d_out["metrics_k%d" % k] = ek.evaluate(M()).reshape(5, -1)
            d_out["ranks_k%d" % k] = lib.ref_topk_rows(Rmask[tus], k)
            # direct call with the dict inputs as well (exercises the row-subset branch, evaluate.py:37-38)
            direct = eval_rating_matrix(Rm.copy(), ek.user_pos_train, ek.user_pos_test, top_k=k)
            assert np.array_equal(direct, d_out["metrics_k%d" % k])
        for key, v in d_out.items():
            cases[name + "/" + key] = v

    run_case("small", 64, 500, 16, [1, 5, 50, 100], (3, 40), (1, 8))
    run_case("subset", 80, 300, 8, [5, 20], (2, 30), (1, 5), test_frac=0.5)          # test users are a strict subset
    run_case("short_truth", 40, 200, 8, [50], (1, 10), (1, 3))                         # |truth| < k for every user
    run_case("neg_inf_tail", 24, 96, 8, [64], (40, 70), (1, 4))                        # I - |train_u| < k for some users
    run_case("c1_shape", 943, 1682, 64, [50], (4, 120), (1, 30), test_frac=0.49, scale=0.3)
    np.savez_compressed(os.path.join(HERE, "eval_cases.npz"), **cases)
    print("eval cases:", sorted({k.split("/")[0] for k in cases}))
    return out


def gen_sampler(utils, data, ctx):
    ds, ev = ctx["ml100k"]
    DataIterator = data.DataIterator
    np.random.seed(2018)                                   # run_model.py:7
    user_pos = utils.csr_to_user_dict(ds.train_matrix)
    all_items = np.arange(ds.train_matrix.shape[1])
    epochs = []
    for ep in range(2):
        users, pos, neg = [], [], []
        for u, p in user_pos.items():                      # model/BPR.py:54-60
            ng = utils.random_choice(all_items, size=len(p), exclusion=p)
            users.extend([u] * len(p)); pos.extend(p.tolist()); neg.extend(ng.tolist())
        it = DataIterator(users, pos, neg, batch_size=512, shuffle=True)
        first = last = None
        nb = 0
        order = []
        for bu, bp, bn in it:
            if first is None:
                first = (bu, bp, bn)
            last = (bu, bp, bn)
            nb += 1
            order.append(np.stack([bu, bp, bn]))
        flat = np.concatenate(order, axis=1)
        epochs.append(dict(neg=np.array(neg, np.uint16), n_batches=nb, last_len=len(last[0]),
                           batched_sha=sha(flat.astype(np.int32)), first_batch=np.array(first, np.int32)))
    print("user0 first negatives:", epochs[0]["neg"][:5])
    assert epochs[0]["neg"][:5].tolist() == [1499, 296, 1538, 608, 825]
    save = {}
    for e, d in enumerate(epochs):
        for k, v in d.items():
            save["epoch%d_%s" % (e, k)] = np.asarray(v)
    # quirk fixture for csr_to_user_dict: empty row, row {0}, row {0,3}, row {2}
    quirk = sp.csr_matrix((np.ones(4), ([1, 2, 2, 3], [0, 0, 3, 2])), shape=(4, 5))
    save["quirk_kept_rows"] = np.array(sorted(utils.csr_to_user_dict(quirk).keys()), np.int32)
    # DNS pool + pick for a small synthetic problem (model/DNSBPR.py:72-79 restated around the
    # reference's random_choice; the logits come from numpy, not from a TF session)
    rng = np.random.default_rng(7)
    U, I, d, neg_num = 30, 120, 8, 15
    P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    rows, cols = [], []
    for u in range(U):
        c = rng.permutation(I)[: int(rng.integers(1, 12))]
        rows += [u] * len(c); cols += c.tolist()
    tr = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, I)); tr.sort_indices()
    up = utils.csr_to_user_dict(tr)
    np.random.seed(99)
    pools, picks, uu, pp = [], [], [], []
    for u, p in up.items():
        logits = (np.matmul(P[u:u + 1], Q.T) + b).reshape(-1)
        pool = utils.random_choice(np.arange(I), size=neg_num * len(p), exclusion=p)
        pool2 = pool.reshape(len(p), neg_num)
        pick = pool2[np.arange(len(p)), np.argmax(logits[pool].reshape(len(p), neg_num), axis=1)]
        pools.append(pool2); picks.append(pick); uu += [u] * len(p); pp += p.tolist()
    save.update(dns_P=P, dns_Q=Q, dns_b=b, dns_indptr=tr.indptr.astype(np.int64), dns_indices=tr.indices.astype(np.int32),
                dns_pool=np.concatenate(pools).astype(np.int32), dns_pick=np.concatenate(picks).astype(np.int32),
                dns_users=np.array(uu, np.int32), dns_pos=np.array(pp, np.int32))
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **save)


def gen_train_steps():
    """Independent implementation: float64 torch autograd of the reference graphs."""
    import torch
    torch.manual_seed(0)
    rng = np.random.default_rng(11)
    save = {}

    def make(U, I, d, B, dup):
        P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32)
        Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
        b = rng.uniform(-0.2, 0.2, I).astype(np.float32)
        if dup:       # force heavy duplication: few distinct rows
            users = rng.integers(0, max(1, U // 8), B); pos = rng.integers(0, max(1, I // 8), B)
            neg = rng.integers(0, max(1, I // 8), B)
        else:
            users = rng.integers(0, U, B); pos = rng.integers(0, I, B); neg = rng.integers(0, I, B)
        return P, Q, b, users.astype(np.int32), pos.astype(np.int32), neg.astype(np.int32)

    def bpr_torch(P, Q, b, u, i, j, lr, reg):
        Pt, Qt, bt = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (P, Q, b)]
        u, i, j = [torch.tensor(x, dtype=torch.long) for x in (u, i, j)]
        yi = (Pt[u] * Qt[i]).sum(1) + bt[i]
        yj = (Pt[u] * Qt[j]).sum(1) + bt[j]
        rank = -torch.nn.functional.logsigmoid(yi - yj + 1e-7).sum()
        l2 = lambda t: 0.5 * (t * t).sum()
        regl = l2(Pt[u]) + l2(Qt[i]) + l2(bt[i]) + l2(Qt[j]) + l2(bt[j])
        (rank + reg * regl).backward()
        new = [(t - lr * t.grad).detach().numpy().astype(np.float32) for t in (Pt, Qt, bt)]
        return new, float(rank), float(regl), [t.grad.numpy() for t in (Pt, Qt, bt)]

    for name, (U, I, d, B, dup) in {"b1": (20, 30, 8, 1, False), "b7": (20, 30, 8, 7, True),
                                    "b128": (60, 50, 20, 128, True), "b512": (943, 1682, 20, 512, False),
                                    "b512dup": (100, 80, 64, 512, True), "d10": (50, 40, 10, 33, False)}.items():
        P, Q, b, u, i, j = make(U, I, d, B, dup)
        new, rank, regl, grads = bpr_torch(P, Q, b, u, i, j, 0.01, 0.05)
        save.update({f"bpr_{name}_P": P, f"bpr_{name}_Q": Q, f"bpr_{name}_b": b, f"bpr_{name}_u": u,
                     f"bpr_{name}_i": i, f"bpr_{name}_j": j, f"bpr_{name}_P1": new[0], f"bpr_{name}_Q1": new[1],
                     f"bpr_{name}_b1": new[2], f"bpr_{name}_loss": np.array([rank, regl])})

    def apr_torch(P, Q, b, aP, aQ, ab, u, i, j, lr, reg, reg_adv, eps):
        T = lambda x, g=True: torch.tensor(x, dtype=torch.float64, requires_grad=g)
        Pt, Qt, bt = T(P), T(Q), T(b)
        ul, il, jl = [torch.tensor(x, dtype=torch.long) for x in (u, i, j)]

        def infer(dP=None, dQ=None, db=None):
            p = Pt[ul] + (dP[ul] if dP is not None else 0)
            qi = Qt[il] + (dQ[il] if dQ is not None else 0)
            qj = Qt[jl] + (dQ[jl] if dQ is not None else 0)
            bi = bt[il] + (db[il] if db is not None else 0)
            bj = bt[jl] + (db[jl] if db is not None else 0)
            return (p * qi).sum(1) + bi, (p * qj).sum(1) + bj

        yi, yj = infer()
        L = torch.nn.functional.softplus(-torch.clamp(yi - yj, -80.0, 1e8)).sum()
        gP, gQ, gb = torch.autograd.grad(L, [Pt, Qt, bt], retain_graph=True)
        nrm = lambda g, dim=None: g * torch.rsqrt(torch.clamp((g * g).sum(dim, keepdim=dim is not None), min=1e-12))
        dP = (nrm(gP, 1) * eps).float().double()
        dQ = (nrm(gQ, 1) * eps).float().double()
        db = (nrm(gb) * eps).float().double()
        yia, yja = infer(dP, dQ, db)
        Ladv = torch.nn.functional.softplus(-torch.clamp(yia - yja, -80.0, 1e8)).sum()
        msq = (Pt[ul] ** 2 + Qt[il] ** 2 + Qt[jl] ** 2 + (bt[il] ** 2)[:, None] + (bt[jl] ** 2)[:, None]).mean()
        opt = L + reg_adv * Ladv + 2 * reg * msq
        G = torch.autograd.grad(opt, [Pt, Qt, bt])
        outs = []
        touched = [np.isin(np.arange(P.shape[0]), u), np.isin(np.arange(Q.shape[0]), np.concatenate([i, j])),
                   np.isin(np.arange(Q.shape[0]), np.concatenate([i, j]))]
        for w, a, g, tm in zip((P, Q, b), (aP, aQ, ab), G, touched):
            g = g.numpy()
            a1 = a.copy(); w1 = w.copy()
            an = (a.astype(np.float64) + g * g).astype(np.float32)
            wn = (w.astype(np.float64) - lr * g / np.sqrt(an.astype(np.float64))).astype(np.float32)
            a1[tm] = an[tm]; w1[tm] = wn[tm]
            outs += [w1, a1]
        return outs, [float(L), float(Ladv), float(msq)], [x.numpy().astype(np.float32) for x in (dP, dQ, db)]

    for name, (U, I, d, B, dup, reg) in {"b1": (20, 30, 8, 1, False, 0.0), "b128": (60, 50, 20, 128, True, 0.0),
                                         "b128reg": (60, 50, 20, 128, True, 0.01),
                                         "b512": (300, 200, 64, 512, False, 0.0)}.items():
        P, Q, b, u, i, j = make(U, I, d, B, dup)
        aP = np.full_like(P, 0.1); aQ = np.full_like(Q, 0.1); ab = np.full_like(b, 0.1)
        outs, losses, deltas = apr_torch(P, Q, b, aP, aQ, ab, u, i, j, 0.05, reg, 1.0, 0.05)
        save.update({f"apr_{name}_P": P, f"apr_{name}_Q": Q, f"apr_{name}_b": b, f"apr_{name}_u": u,
                     f"apr_{name}_i": i, f"apr_{name}_j": j, f"apr_{name}_reg": np.array(reg),
                     f"apr_{name}_P1": outs[0], f"apr_{name}_aP1": outs[1], f"apr_{name}_Q1": outs[2],
                     f"apr_{name}_aQ1": outs[3], f"apr_{name}_b1": outs[4], f"apr_{name}_ab1": outs[5],
                     f"apr_{name}_loss": np.array(losses), f"apr_{name}_dP": deltas[0], f"apr_{name}_dQ": deltas[1],
                     f"apr_{name}_db": deltas[2]})
    np.savez_compressed(os.path.join(HERE, "train_steps.npz"), **save)
    print("train-step cases:", len(save))


def gen_splitter():
    """Splitter vectors -> the reference's own data/utils.py functions (load_data, filter_data, remap_id,
    split_by_ratio, split_by_loo: data/utils.py:7-98) on a small synthetic UIRT file with sparse raw ids,
    rare items, short users and repeated timestamps; the random splits seeded through numpy's legacy generator."""
    from data.utils import filter_data, load_data, remap_id, split_by_loo, split_by_ratio
    rng = np.random.default_rng(31)
    lines = []
    for u in range(40):
        n = int(rng.integers(1, 15))
        for it in rng.choice(60, n, replace=False):
            lines.append((100 + 3 * u, 1000 + 7 * int(it), int(rng.integers(1, 6)), 900000 + int(rng.integers(0, 12))))
    raw = np.array([lines[i] for i in rng.permutation(len(lines))], dtype=np.int64)
    work = os.path.join(ROOT, "build")            # scratch, not tracked
    os.makedirs(work, exist_ok=True)
    path = os.path.join(work, "splitter_input.txt")
    np.savetxt(path, raw, fmt="%d", delimiter=",")
    save = {"raw": raw}
    cols = ["user", "item", "rating", "time"]
    table = lambda a: np.stack([a[c] for c in cols], axis=1).astype(np.int64) if len(a) else np.zeros((0, 4), np.int64)
    for name, (umin, imin) in {"nofilter": (0, 0), "filtered": (3, 2)}.items():
        data = remap_id(filter_data(load_data(path, sep=",", columns=cols), user_min=umin, item_min=imin))
        remapped, user2id, item2id = data
        save["%s_users" % name] = np.array(list(user2id.keys()), np.int64)
        save["%s_items" % name] = np.array(list(item2id.keys()), np.int64)
        save["%s_remapped" % name] = table(remapped.copy())
        for by_time in (True, False):
            np.random.seed(7)
            tr, te = split_by_ratio(remapped.copy(), ratio=0.8, by_time=by_time)
            save["%s_ratio_%d_train" % (name, by_time)], save["%s_ratio_%d_test" % (name, by_time)] = table(tr), table(te)
            np.random.seed(7)
            tr, te = split_by_loo(remapped.copy(), by_time=by_time)
            save["%s_loo_%d_train" % (name, by_time)], save["%s_loo_%d_test" % (name, by_time)] = table(tr), table(te)
    np.savez_compressed(os.path.join(HERE, "splitter.npz"), **save)
    print("splitter cases:", len(save))


if __name__ == "__main__":
    if "--only-splitter" in sys.argv:
        sys.path.insert(0, REF)
        gen_splitter()
        sys.exit(0)
    import oracle as orc  # noqa: E402
    os.chdir(REF)        # data/dataset.py paths in tfrec.ini are relative to the tree
    utils, data, evaluator = import_reference()
    ctx = gen_eval(utils, data, evaluator, orc)
    gen_sampler(utils, data, ctx)
    gen_train_steps()
    gen_splitter()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
